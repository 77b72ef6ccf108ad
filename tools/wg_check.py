"""wg_check.py — GPU check + timing of the generated 3x3 weight-gradient kernels through the per-op C-ABI (exact integer data)."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from sota_imagenet_amd import ops  # noqa: E402
sys.path.insert(0, "tests")
from test_dconv_gpu import WG_SHAPES, _wgrad_ref  # noqa: E402

dev = "cuda"
BIG = bool(os.environ.get("WG_BIG_ONLY"))  # only the batch-256 shapes (counter passes average over a kernel's launches)
K1 = [(256, 14, 1024, 256), (256, 14, 256, 1024), (256, 7, 2048, 512), (256, 7, 512, 2048), (256, 28, 512, 128), (256, 28, 128, 512), (256, 56, 256, 64),
      (256, 56, 64, 256), (256, 56, 256, 128), (256, 28, 512, 256), (256, 14, 1024, 512), (3, 14, 256, 1024), (5, 7, 2048, 512), (3, 56, 64, 256)]
for (N, H, Cin, Cout) in K1:
    if BIG and N != 256:
        continue
    torch.manual_seed(0)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    dy = torch.randint(-2, 3, (N, H, H, Cout), device=dev).to(torch.bfloat16)
    ref = (dy.float().reshape(-1, Cout).t() @ x.float().reshape(-1, Cin)).reshape(Cout, 1, 1, Cin)
    dw = ops.conv2d_wgrad(dy, x, 1, 1, 1, 0)
    torch.cuda.synchronize()
    ok = torch.equal(dw, ref)
    for _ in range(3):
        ops.conv2d_wgrad(dy, x, 1, 1, 1, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.conv2d_wgrad(dy, x, 1, 1, 1, 0)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    fl = 2.0 * N * H * H * Cin * Cout
    print(f"1x1 N={N} H={H} {Cin}->{Cout}: exact={ok} maxerr={(dw - ref).abs().max().item():.3g}  {us:.1f} us (wgrad + reduce)  {fl / us / 1e6:.0f} TF/s", flush=True)
for (N, H, Cin, Cout) in WG_SHAPES + [(256, 112, 64, 64)]:
    if BIG and N != 256:
        continue
    torch.manual_seed(0)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    dy = torch.randint(-2, 3, (N, H, H, Cout), device=dev).to(torch.bfloat16)
    ref = _wgrad_ref(dy, x)
    dw = ops.conv2d_wgrad(dy, x, 3, 3, 1, 1)
    torch.cuda.synchronize()
    ok = torch.equal(dw, ref)
    for _ in range(3):
        ops.conv2d_wgrad(dy, x, 3, 3, 1, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.conv2d_wgrad(dy, x, 3, 3, 1, 1)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f"N={N} H={H} {Cin}->{Cout}: exact={ok} maxerr={(dw - ref).abs().max().item():.3g}  {us:.1f} us (wgrad + reduce)  {fl / us / 1e6:.0f} TF/s", flush=True)
