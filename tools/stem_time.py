"""On the GPU box: time the bf16 stem forward (per-op C-ABI, no statistics) at batch N / S px: tools/stem_time.py [N] [S]"""
import sys, time, torch
sys.path.insert(0, '.')
from sota_imagenet_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 224
x = torch.randn(N, 3, S, S, device='cuda')
w = torch.randn(64, 7, 7, 3, device='cuda') * 0.1
xpad = ops.stem_ingest(x, torch.bfloat16)
for _ in range(3): y = ops.stem_fwd(xpad, w, N, S, S, torch.bfloat16)
big = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
ts = []
for _ in range(10):
    big.zero_()  # cold caches
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); y = ops.stem_fwd(xpad, w, N, S, S, torch.bfloat16); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(f"stem fwd N={N} S={S}: median {ts[len(ts)//2]:.1f} us (incl. the 6 us weight pack), min {ts[0]:.1f}")
