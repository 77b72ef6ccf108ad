"""GPU check + timing of the generated direct-conv kernels through the per-op C-ABI.
  python tools/dconv_check.py exact     integer data: forward (+ BN statistics rows) and dgrad against torch, bit for bit
  python tools/dconv_check.py time      per-launch time of the layer-3 / layer-4 3x3 shapes (run again with MI355_DCONV=0 for the A/B)
"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from sota_imagenet_amd import ops  # noqa: E402

dt = torch.bfloat16
SHAPES = [(256, 14, 256, 256, 3), (256, 7, 512, 512, 3), (256, 28, 128, 128, 3), (256, 56, 64, 64, 3), (256, 14, 256, 1024, 1), (256, 14, 1024, 256, 1),
          (256, 7, 2048, 512, 1), (256, 7, 512, 2048, 1), (256, 28, 512, 256, 1), (256, 14, 1024, 512, 1), (256, 28, 512, 128, 1)]


def exact():
    torch.manual_seed(0)
    ok = True
    for (N, H, Cin, Cout, K) in SHAPES:
        x = torch.randint(-2, 3, (N, H, H, Cin), device="cuda").to(dt)
        w = torch.randint(-2, 3, (Cout, K, K, Cin), device="cuda").to(dt)
        y, part = ops.conv2d_fwd(x, w, 1, K // 2, stats=True)
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1).to(dt)
        e = (y.float() - ref.float()).abs().max().item()
        s1 = ref.float().sum(dim=(0, 1, 2))
        s2 = (ref.float() ** 2).sum(dim=(0, 1, 2))
        es = None
        if part is not None:
            es = max((part[:, 0].double().sum(0) - s1.double()).abs().max().item() / s1.abs().max().item(),
                     (part[:, 1].double().sum(0) - s2.double()).abs().max().item() / s2.abs().max().item())
        dy = torch.randint(-2, 3, (N, H, H, Cout), device="cuda").to(dt)
        dx = ops.conv2d_dgrad(dy, w, (N, H, H, Cin), 1, K // 2)
        refd = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1).to(dt)
        ed = (dx.float() - refd.float()).abs().max().item()
        print(f"{N}x{H}x{H} {Cin}->{Cout} k{K}: fwd max err {e}  stats rows {None if part is None else part.shape[0]} rel err {es}  dgrad max err {ed}", flush=True)
        ok = ok and e == 0 and ed == 0 and (es is None or es < 1e-6)
    print("EXACT OK" if ok else "EXACT FAILED")
    return ok


def timing():
    for (N, H, Cin, Cout, K) in SHAPES:
        # a pool of input tensors larger than the 256 MiB Infinity Cache, so every launch starts from cold caches
        pool = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(12)]
        w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(dt)
        for stats in (False, True):
            for i in range(6):
                ops.conv2d_fwd(pool[i % 12], w, 1, K // 2, stats=stats)
            torch.cuda.synchronize()
            evs = []
            for i in range(36):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.conv2d_fwd(pool[i % 12], w, 1, K // 2, stats=stats)
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            med = ts[len(ts) // 2] * 1e-3
            fl = 2.0 * N * H * H * Cout * Cin * K * K
            print(f"fwd {N}x{H}x{H} {Cin}->{Cout} k{K} stats={int(stats)}: median {med*1e6:7.1f} us  min {ts[0]*1e3:7.1f} us  {fl/med/1e12:7.1f} TF/s  (MI355_DCONV={os.environ.get('MI355_DCONV', '1')})", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
    if mode == "exact":
        sys.exit(0 if exact() else 1)
    timing()
