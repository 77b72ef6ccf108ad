"""times the 3x3 / stride-1 max pool of BResNet-50's stem (256 x 112 x 112 x 64, bf16) forward and backward; MI355_POOL_SEG=1: the point-wise kernels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sota_imagenet_amd import ops
x = torch.randn(256, 112, 112, 64, device="cuda").bfloat16()
dy = torch.randn_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
y, idx = ops.maxpool3s1_fwd(x)
print("MI355_POOL_SEG=%s fwd %.0f us  bwd %.0f us" % (os.environ.get("MI355_POOL_SEG", "-"), t(lambda: ops.maxpool3s1_fwd(x)), t(lambda: ops.maxpool3s1_bwd(dy, idx))))
