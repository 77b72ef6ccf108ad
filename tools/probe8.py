"""igemm8 with the operand loads turned into zero-record reads (MI355_IGEMM8_DBG=3: no L2/HBM traffic on the load side):
what remains is MFMA + epilogue + stores.  python tools/probe8.py"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from sota_imagenet_amd import native, ops
dt = torch.bfloat16
N = 256
for (name, H, Cin, Cout, K, tile) in [("l3.c3 fwd", 14, 256, 1024, 1, "224x256"), ("l3.c1 fwd", 14, 1024, 256, 1, "224x256"), ("l2.c3 fwd", 28, 128, 512, 1, "224x256"),
                                     ("l2.c3 fwd", 28, 128, 512, 1, "256x128f"), ("l3.c2 fwd", 14, 256, 256, 3, "224x256"), ("l4.c3 fwd", 7, 512, 2048, 1, "224x256")]:
    M = N * H * H
    nset = max(3, int(1.6e9 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
    w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(dt)
    os.environ["MI355_IGEMM8"] = tile
    native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
    out = []
    for dbg in ("0", "1", "2", "3"):
        os.environ["MI355_IGEMM8_DBG"] = dbg
        for i in range(nset):
            ops.conv2d_fwd(xs[i], w, 1, K // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for i in range(nset):
                ops.conv2d_fwd(xs[i], w, 1, K // 2)
        torch.cuda.synchronize()
        out.append(f"dbg{dbg} {(time.perf_counter() - t0) / (3 * nset) * 1e6:6.1f}us")
    fl = 2.0 * M * Cin * Cout * K * K
    print(f"{name:10s} {tile:9s} GF {fl/1e9:6.1f} | " + " | ".join(out), flush=True)
    del xs

# the 4-wave kernel (conv_igemm.hip): MI355_IGEMM_DBG 1 = no epilogue at all, 2 = full epilogue but its stores hit one trash page
# (compiled in only by `make -C sota_imagenet_amd/csrc probes`; run this script with MI355RN_LIB=sota_imagenet_amd/lib/variant_probes.so)
os.environ["MI355_IGEMM8"] = "0"
native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
os.environ.pop("MI355_IGEMM8_DBG", None)
for (name, H, Cin, Cout, K) in [("l1.c3 fwd", 56, 64, 256, 1), ("l1.c1 fwd", 56, 256, 64, 1), ("l2.c3 fwd", 28, 128, 512, 1), ("l2.c1 fwd", 28, 512, 128, 1),
                                ("l3.c3 fwd", 14, 256, 1024, 1), ("l1.c2 fwd", 56, 64, 64, 3), ("l2.c2 fwd", 28, 128, 128, 3)]:
    M = N * H * H
    nset = max(3, int(1.6e9 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
    w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(dt)
    out = []
    for dbg in ("0", "2", "1"):
        os.environ["MI355_IGEMM_DBG"] = dbg
        for i in range(nset):
            ops.conv2d_fwd(xs[i], w, 1, K // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for i in range(nset):
                ops.conv2d_fwd(xs[i], w, 1, K // 2)
        torch.cuda.synchronize()
        out.append(f"dbg{dbg} {(time.perf_counter() - t0) / (3 * nset) * 1e6:6.1f}us")
    os.environ.pop("MI355_IGEMM_DBG", None)
    print(f"4-wave {name:10s} | " + " | ".join(out), flush=True)
    del xs

# dgrad with the shortcut addend: how much of its cost is the addend's HBM latency inside the epilogue (probe 4 reads it from a cached page)
for (name, H, Cin, Cout) in [("l3.c1 dgrad", 14, 1024, 256), ("l2.c1 dgrad", 28, 512, 128), ("l4.c1 dgrad", 7, 2048, 512)]:
    M = N * H * H
    nset = max(3, int(1.6e9 // (M * (2 * Cin + Cout) * 2)) + 1)
    dys = [torch.randn(N, H, H, Cout, device="cuda").to(dt) for _ in range(nset)]
    adds = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05).to(dt)
    out = []
    for label, dbg, use_add in (("plain", "0", False), ("+addend", "0", True), ("+addend cached", "4", True), ("+addend, no stores", "2", True), ("no epilogue", "1", True)):
        os.environ["MI355_IGEMM_DBG"] = dbg
        f = lambda i: ops.conv2d_dgrad(dys[i], w, (N, H, H, Cin), 1, 0, addend=adds[i] if use_add else None)
        for i in range(nset):
            f(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for i in range(nset):
                f(i)
        torch.cuda.synchronize()
        out.append(f"{label} {(time.perf_counter() - t0) / (3 * nset) * 1e6:6.1f}us")
    os.environ.pop("MI355_IGEMM_DBG", None)
    print(f"4-wave {name:12s} | " + " | ".join(out), flush=True)
    del dys, adds

# weight gradients: MI355_WGRAD_DBG=3 replaces both operand loads by zero-record reads (no L2 / HBM traffic)
os.environ.pop("MI355_IGEMM8", None)
native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
for (name, H, Cin, Cout, K) in [("l3.c2.w", 14, 256, 256, 3), ("l3.c1.w", 14, 1024, 256, 1), ("l3.c3.w", 14, 256, 1024, 1), ("l4.c2.w", 7, 512, 512, 3),
                                ("l2.c2.w", 28, 128, 128, 3), ("l2.c1.w", 28, 512, 128, 1), ("l1.c3.w", 56, 64, 256, 1), ("l1.c2.w", 56, 64, 64, 3)]:
    M = N * H * H
    nset = max(3, int(1.6e9 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
    dys = [torch.randn(N, H, H, Cout, device="cuda").to(dt) for _ in range(nset)]
    out = []
    for dbg in ("0", "3"):
        os.environ["MI355_WGRAD_DBG"] = dbg
        f = lambda i: ops.conv2d_wgrad(dys[i], xs[i], K, K, 1, K // 2)
        for i in range(nset):
            f(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for i in range(nset):
                f(i)
        torch.cuda.synchronize()
        out.append(f"dbg{dbg} {(time.perf_counter() - t0) / (3 * nset) * 1e6:6.1f}us")
    os.environ.pop("MI355_WGRAD_DBG", None)
    fl = 2.0 * M * Cin * Cout * K * K
    print(f"wgrad {name:8s} GF {fl / 1e9:5.1f} MB {M * (Cin + Cout) * 2 / 1e6:5.0f} | " + " | ".join(out), flush=True)
    del xs, dys
