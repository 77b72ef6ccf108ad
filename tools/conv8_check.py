"""A/B of the 8-wave igemm kernel (conv_igemm8.hip) against the 4-wave one through the per-op C-ABI, in ONE process.

  python tools/conv8_check.py exact    small + ragged shapes, integer data: forced tiles must equal the old kernel bit for bit
  python tools/conv8_check.py time     the network's long-reduction layer shapes at batch 256: us and TF/s per tile choice
MI355_IGEMM8 is re-read by the library at every launch, so the variants alternate inside the process (same device, same clocks).
"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from sota_imagenet_amd import native, ops  # noqa: E402

dt = torch.bfloat16
TILES = ["256x256", "224x256", "256x128", "224x128"]


def setv(v):
    if v is None:
        os.environ["MI355_IGEMM8"] = "0"
        native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
    else:
        os.environ["MI355_IGEMM8"] = v
        native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)


def run(kind, x, w, dy, add, shape, s, pad):
    if kind == "fwd":
        return ops.conv2d_fwd(x, w, s, pad)
    if kind == "dgrad":
        return ops.conv2d_dgrad(dy, w, shape, s, pad)
    return ops.conv2d_dgrad(dy, w, shape, s, pad, addend=add)


def exact():
    cases = [(2, 14, 14, 128, 256, 3, 1), (3, 7, 7, 256, 128, 3, 1), (4, 14, 14, 256, 256, 1, 1), (5, 9, 9, 128, 256, 3, 2),
             (2, 8, 8, 128, 256, 1, 2), (16, 14, 14, 256, 512, 3, 1), (40, 14, 14, 128, 128, 3, 1), (64, 14, 14, 192, 256, 3, 2),
             (33, 7, 7, 512, 256, 1, 1)]
    bad = 0
    g = torch.Generator().manual_seed(3)
    for (N, H, W, Cin, Cout, K, s) in cases:
        pad = K // 2
        Ho = (H + 2 * pad - K) // s + 1
        x = torch.randint(-2, 3, (N, H, W, Cin), generator=g).float().cuda().to(dt)
        w = torch.randint(-2, 3, (Cout, K, K, Cin), generator=g).float().cuda().to(dt)
        dy = torch.randint(-2, 3, (N, Ho, Ho, Cout), generator=g).float().cuda().to(dt)
        add = torch.randint(-3, 4, (N, H, W, Cin), generator=g).float().cuda().to(dt)
        for kind in ("fwd", "dgrad", "dgrad+add"):
            if kind != "fwd" and s == 2 and H % 2:
                continue
            setv(None)
            ref = run(kind, x, w, dy, add, (N, H, W, Cin), s, pad)
            torch.cuda.synchronize()
            for tile in TILES:
                for ko in ("", "k", "f", "kf"):
                    ncols = Cout if kind == "fwd" else Cin
                    if ncols % int(tile.split("x")[1]):
                        continue
                    setv(tile + ko)
                    got = run(kind, x, w, dy, add, (N, H, W, Cin), s, pad)
                    torch.cuda.synchronize()
                    ok = torch.equal(got, ref)
                    if not ok:
                        bad += 1
                        d = (got.float() - ref.float()).abs()
                        nz = (d > 0).nonzero()
                        print(f"MISMATCH {kind} {tile}{ko} case {(N, H, W, Cin, Cout, K, s)}: {int((d > 0).sum())} of {d.numel()} differ, max {d.max().item()}, first {nz[0].tolist()} last {nz[-1].tolist()}", flush=True)
        print("case", (N, H, W, Cin, Cout, K, s), "done", flush=True)
    print("EXACT", "FAILED" if bad else "OK", bad)
    return bad


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def timing():
    shapes = [("l2.c2", 256, 28, 128, 128, 3, 1), ("l3.c2", 256, 14, 256, 256, 3, 1), ("l4.c2", 256, 7, 512, 512, 3, 1),
              ("l3.c1", 256, 14, 1024, 256, 1, 1), ("l3.c3", 256, 14, 256, 1024, 1, 1), ("l4.c1", 256, 7, 2048, 512, 1, 1),
              ("l4.c3", 256, 7, 512, 2048, 1, 1), ("l3.0.c2", 256, 28, 256, 256, 3, 2), ("l4.0.ds", 256, 14, 1024, 2048, 1, 2),
              ("l2.c1", 256, 28, 512, 128, 1, 1), ("l2.c3", 256, 28, 128, 512, 1, 1)]
    for (name, N, H, Cin, Cout, K, s) in shapes:
        pad = K // 2
        Ho = (H + 2 * pad - K) // s + 1
        x = torch.randn(N, H, H, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(dt)
        dy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(dt)
        fl = 2.0 * N * Ho * Ho * Cout * Cin * K * K
        for kind in ("fwd", "dgrad"):
            ncols = Cout if kind == "fwd" else Cin
            res = []
            setv(None)
            ref = run(kind, x, w, dy, None, (N, H, H, Cin), s, pad)
            variants = [None] + [t + ko for t in TILES for ko in ("", "f", "kf") if ncols % int(t.split("x")[1]) == 0 and not ("k" in ko and K == 1 and kind == "fwd")]
            # two interleaved rounds per variant
            best = {}
            for rnd in range(2):
                for v in variants:
                    setv(v)
                    t = bench(lambda: run(kind, x, w, dy, None, (N, H, H, Cin), s, pad))
                    best[v] = min(best.get(v, 1e9), t)
            for v in variants:
                setv(v)
                got = run(kind, x, w, dy, None, (N, H, H, Cin), s, pad)
                err = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
                res.append(f"{v or 'old':9s} {best[v] * 1e6:6.1f}us {fl / best[v] / 1e12:5.0f}TF" + ("" if err < 1e-3 else f" ERR{err:.0e}"))
            if Cin % 128 == 0 and Cout % 128 == 0:  # fp8 operand form of the same 8-wave kernel (quantisation not timed)
                xq, wq, dyq = ops.quantize_fp8(x, 16.0), ops.quantize_fp8(w, 256.0), ops.quantize_fp8(dy, 16.0)
                wt8 = wq.view(torch.uint8).permute(3, 1, 2, 0).contiguous()
                f8 = (lambda: ops.conv2d_fwd_fp8(xq, wq, s, pad)) if kind == "fwd" else (lambda: ops.conv2d_dgrad_fp8(dyq, wq, (N, H, H, Cin), s, pad, wt=wt8))
                t8 = min(bench(f8), bench(f8))
                res.append(f"fp8 {t8 * 1e6:6.1f}us {fl / t8 / 1e12:5.0f}TF")
            print(f"{name:8s} {kind:5s} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
    if mode == "exact":
        sys.exit(1 if exact() else 0)
    timing()
