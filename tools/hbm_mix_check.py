"""HBM streaming yardsticks by read:write mix under cold caches (buffers rotate through > 1.5 GB): fill (0R:1W), copy (1R:1W),
add (2R:1W), sum-reduce (1R:0W) — the write-heavy convs (conv3 forward: 1 part read, 4 parts write) must be judged against the
write-heavy yardsticks.  python tools/hbm_mix_check.py"""
import time
import torch

dt = torch.bfloat16
n = 205_520_896  # elements: 411 MB bf16 = layer-1's 256-channel tensor at batch 256
sets = 5
a = [torch.empty(n, device="cuda", dtype=dt).normal_() for _ in range(sets)]
b = [torch.empty(n, device="cuda", dtype=dt).normal_() for _ in range(sets)]
c = [torch.empty(n, device="cuda", dtype=dt) for _ in range(sets)]
q = [torch.empty(n // 4, device="cuda", dtype=dt).normal_() for _ in range(sets)]


def t(fn, reps=4):
    for i in range(sets):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(sets):
            fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * sets)


B = n * 2
for name, fn, by in [("fill   0R:1W", lambda i: c[i].fill_(1.0), B), ("copy   1R:1W", lambda i: c[i].copy_(a[i]), 2 * B), ("add    2R:1W", lambda i: torch.add(a[i], b[i], out=c[i]), 3 * B),
                     ("relu_  1R:1W in place", lambda i: a[i].relu_(), 2 * B), ("sum    1R:0W", lambda i: a[i].sum(), B),
                     ("expand 1R:4W (quarter-size source repeated)", lambda i: c[i].view(4, -1).copy_(q[i].view(1, -1).expand(4, -1)), B + B // 4)]:
    s = t(fn)
    print(f"{name:48s} {s * 1e6:7.1f} us  {by / s / 1e12:5.2f} TB/s", flush=True)
