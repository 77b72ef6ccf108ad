"""s2_time.py — the three 3x3 / stride-2 convolutions of ResNet-50 (conv2 of layer2.0 / 3.0 / 4.0) at batch 256 through the per-op C-ABI, generated kernels
(MI355_DCONV_S2=1) against the implicit-GEMM kernels (=0), same process, alternating; a 512 MB buffer is rewritten between launches (cold caches).
   python tools/s2_time.py [dgrad|fwd|wgrad ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sota_imagenet_amd import native, ops  # noqa: E402

dt = torch.bfloat16
what = sys.argv[1:] or ["dgrad", "fwd", "wgrad"]
flush = torch.empty(256 << 20, dtype=torch.bfloat16, device="cuda")


def timed(fn, reps=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(reps):
        flush.add_(1)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def knob(v):
    os.environ["MI355_DCONV_S2"] = str(v)
    native.lib().mi355_reload_knobs()


for (N, H, C) in [(256, 56, 128), (256, 28, 256), (256, 14, 512)]:
    Ho = H // 2
    x = torch.randn(N, H, H, C, device="cuda").to(dt)
    w = (torch.randn(C, 3, 3, C, device="cuda") * 0.05).to(dt)
    dy = torch.randn(N, Ho, Ho, C, device="cuda").to(dt)
    y = torch.randn(N, H, H, C, device="cuda").to(dt)
    bits = torch.randint(0, 256, (N, H, H, C // 8), device="cuda", dtype=torch.uint8)
    mean, invstd = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5
    fl = 2.0 * N * Ho * Ho * C * C * 9
    for kind in what:
        row = []
        for v in (0, 1, 0, 1):
            knob(v)
            if kind == "dgrad":
                f = lambda: ops.conv2d_dgrad_bn(dy, w, (N, H, H, C), 2, 1, bn_y=y, bn_bits=bits, bn_mean=mean, bn_invstd=invstd)
            elif kind == "fwd":
                f = lambda: ops.conv2d_fwd(x, w, 2, 1, stats=True)
            else:
                f = lambda: ops.conv2d_wgrad(dy, x, 3, 3, 2, 1)
            f()
            name = ops.last_conv_kernel()
            t = timed(f)
            row.append("S2=%d %-22s %7.1f us %6.0f TF/s" % (v, name, t, fl / t / 1e6))
        print("%-5s %3d x %3d x %3d -> %3d: " % (kind, H, H, C, Ho) + " | ".join(row), flush=True)
knob(1)
