"""conv3x3.hip (direct 3x3, 64 channels) against the implicit-GEMM kernel through the per-op C-ABI, one process.
  python tools/conv3_check.py exact   integer data: must equal the igemm kernel bit for bit (forward, dgrad, dgrad + addend)
  python tools/conv3_check.py time    layer-1 shape at batch 256: us and TF/s, both kernels
MI355_CONV3 is re-read by the library at every launch."""
import os, sys, time, torch
sys.path.insert(0, ".")
from sota_imagenet_amd import ops
dt = torch.bfloat16

def run(kind, x, w, dy, add, s=1, pad=1):
    if kind == "fwd": return ops.conv2d_fwd(x, w, s, pad)
    if kind == "dgrad": return ops.conv2d_dgrad(dy, w, tuple(x.shape), s, pad)
    return ops.conv2d_dgrad(dy, w, tuple(x.shape), s, pad, addend=add)

def exact():
    g = torch.Generator().manual_seed(3)
    bad = 0
    for (N, H, W) in [(2, 56, 56), (3, 8, 8), (1, 16, 24), (5, 40, 40), (2, 28, 28), (7, 14, 14), (2, 80, 80), (3, 7, 7), (300, 8, 8)]:
        x = torch.randint(-2, 3, (N, H, W, 64), generator=g).float().cuda().to(dt)
        w = torch.randint(-2, 3, (64, 3, 3, 64), generator=g).float().cuda().to(dt)
        dy = torch.randint(-2, 3, (N, H, W, 64), generator=g).float().cuda().to(dt)
        add = torch.randint(-3, 4, (N, H, W, 64), generator=g).float().cuda().to(dt)
        for kind in ("fwd", "dgrad", "dgrad+add"):
            os.environ["MI355_CONV3"] = "0"
            ref = run(kind, x, w, dy, add)
            os.environ["MI355_CONV3"] = "1"
            got = run(kind, x, w, dy, add)
            torch.cuda.synchronize()
            ok = torch.equal(ref, got)
            if not ok:
                bad += 1
                d = (ref.float() - got.float()).abs()
                idx = torch.nonzero(d > 0)
                print("MISMATCH", (N, H, W), kind, "count", idx.shape[0], "of", d.numel(), "first", idx[:5].tolist(), "max", d.max().item())
            else:
                print("ok", (N, H, W), kind)
    print("bad", bad)
    return bad

def timeit():
    N, H = 256, 56
    x = torch.randn(N, H, H, 64, device="cuda").to(dt); w = (torch.randn(64, 3, 3, 64, device="cuda") * 0.05).to(dt)
    dy = torch.randn(N, H, H, 64, device="cuda").to(dt); add = torch.randn(N, H, H, 64, device="cuda").to(dt)
    big = torch.empty(1 << 28, device="cuda")  # 1 GiB: flushes L2 / MALL between launches
    fl = 2.0 * N * H * H * 64 * 64 * 9
    for kind in (("fwd",) if os.environ.get("MI355_CONV3_DBG") else ("fwd", "dgrad", "dgrad+add")):
        for env in ("0", "1"):
            os.environ["MI355_CONV3"] = env
            for _ in range(3): run(kind, x, w, dy, add)
            ts = []
            for _ in range(10):
                big.zero_(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(kind, x, w, dy, add); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort(); t = ts[len(ts) // 2]
            print(f"{kind:10s} conv3={env}: {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s (cold caches, median of 10)")

if __name__ == "__main__":
    sys.exit(exact() if sys.argv[1:] == ["exact"] else timeit())
