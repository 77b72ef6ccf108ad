"""probe: the stride-2 1x1 downsample forward of the three striding blocks — as launched today against a stride-1 launch on a compacted input"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sota_imagenet_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (H, Cin, Cout) in ((56, 256, 512), (28, 512, 1024), (14, 1024, 2048)):
    x = torch.randn(256, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05).bfloat16()
    y0, p0 = ops.conv2d_fwd(x, w, 2, 0, stats=True)
    k0 = ops.last_conv_kernel()
    xc = x[:, ::2, ::2].contiguous()
    y1, p1 = ops.conv2d_fwd(xc, w, 1, 0, stats=True)
    k1 = ops.last_conv_kernel()
    print(H, Cin, Cout, "same bits:", torch.equal(y0, y1), "| strided %s %.1f us | compact %s %.1f us | torch gather %.1f us" % (
        k0, t(lambda: ops.conv2d_fwd(x, w, 2, 0, stats=True)), k1, t(lambda: ops.conv2d_fwd(xc, w, 1, 0, stats=True)), t(lambda: x[:, ::2, ::2].contiguous())))
