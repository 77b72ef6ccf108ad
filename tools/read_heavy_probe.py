"""probe: why does a read-heavy 1x1 launch (256 -> 64 at 56 x 56, batch 256) take 130 us in the step and 100 us in the harness?
times ops.conv2d_fwd on (a) random data, (b) post-ReLU-like data (half zeros), (c) data a previous elementwise kernel has just written,
(d) a rotating pool of inputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sota_imagenet_amd import ops

def t(fn, pre=None, n=30):
    ts = []
    for i in range(n + 3):
        if pre: pre(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); b.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

N, H, Cin, Cout = 256, 56, 256, 64
w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05).bfloat16()
pool = [torch.randn(N, H, H, Cin, device="cuda").bfloat16() for _ in range(4)]
relu = [p.clamp_min(0) for p in pool]
print("kernel:", (ops.conv2d_fwd(pool[0], w, 1, 0, stats=True), ops.last_conv_kernel())[1])
print("random, pool of 4      %.1f us" % t(lambda i: ops.conv2d_fwd(pool[i % 4], w, 1, 0, stats=True)))
print("half zeros, pool of 4  %.1f us" % t(lambda i: ops.conv2d_fwd(relu[i % 4], w, 1, 0, stats=True)))
print("same tensor every time %.1f us" % t(lambda i: ops.conv2d_fwd(pool[0], w, 1, 0, stats=True)))
print("just written (relu_)   %.1f us" % t(lambda i: ops.conv2d_fwd(pool[i % 4], w, 1, 0, stats=True), pre=lambda i: pool[i % 4].add_(1.0)))
big = torch.empty(40 * (1 << 30), dtype=torch.uint8, device="cuda")  # one large arena: tensors 2.5 GB apart
views = [big[k * (5 << 29): k * (5 << 29) + N * H * H * Cin * 2].view(torch.bfloat16).view(N, H, H, Cin) for k in range(8)]
for v in views: v.copy_(pool[0])
print("views of a 40 GB arena %.1f us" % t(lambda i: ops.conv2d_fwd(views[i % 8], w, 1, 0, stats=True)))
# (e) behind a matrix-pipe-heavy kernel (power / clocks): a 8192^3 bf16 GEMM (~1.1 PFLOP -> ~1 ms) before every timed launch
A = torch.randn(8192, 8192, device="cuda").bfloat16(); B = torch.randn(8192, 8192, device="cuda").bfloat16()
print("behind a large GEMM    %.1f us" % t(lambda i: ops.conv2d_fwd(pool[i % 4], w, 1, 0, stats=True), pre=lambda i: torch.matmul(A, B)))
def heavy(i):
    for _ in range(8): torch.matmul(A, B)
print("behind 8 large GEMMs   %.1f us" % t(lambda i: ops.conv2d_fwd(pool[i % 4], w, 1, 0, stats=True), pre=heavy))
# (f) behind an elementwise kernel with the traffic of bn_apply + residual (2 reads + 1 write of 411 MB)
o = torch.empty_like(pool[0])
print("behind add(out=)       %.1f us" % t(lambda i: ops.conv2d_fwd(o, w, 1, 0, stats=True), pre=lambda i: torch.add(pool[i % 4], pool[(i + 1) % 4], out=o)))
