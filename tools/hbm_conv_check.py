"""HBM-bound convs under cold caches: the layer-1/2 1x1 shapes of the step, each call on a fresh set of buffers (the sets rotate
through > 1.5 GB so neither L2 nor the 256 MB MALL holds a tensor from the previous call — the condition inside the train step).
Prints us and algorithmic TB/s for the rule-selected kernel, forced 8-wave tiles, and a torch elementwise kernel moving the same
number of bytes (what a pure streaming kernel reaches on this box).
  python tools/hbm_conv_check.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from sota_imagenet_amd import native, ops  # noqa: E402

dt = torch.bfloat16
N = 256
# name, H, Cin, Cout, kind
SHAPES = [("l1.c1 fwd", 56, 256, 64, "fwd"), ("l1.c3 fwd", 56, 64, 256, "fwd"), ("l1.c3 dgrad", 56, 64, 256, "dgrad"),
          ("l1.c1 dgrad+add", 56, 256, 64, "dgrad+add"), ("l2.c1 fwd", 28, 512, 128, "fwd"), ("l2.c3 fwd", 28, 128, 512, "fwd"),
          ("l2.c3 dgrad", 28, 128, 512, "dgrad"), ("l2.c1 dgrad+add", 28, 512, 128, "dgrad+add"), ("l3.c3 fwd", 14, 256, 1024, "fwd"),
          ("l3.c1 fwd", 14, 1024, 256, "fwd"), ("l3.c1 dgrad+add", 14, 1024, 256, "dgrad+add")]
VARIANTS = [None, "0"]


def main():
    for (name, H, Cin, Cout, kind) in SHAPES:
        M = N * H * H
        by = {"fwd": M * (Cin + Cout) * 2, "dgrad": M * (Cin + Cout) * 2, "dgrad+add": M * (Cout + 2 * Cin) * 2}[kind]
        nset = max(3, int(1.6e9 // by) + 1)
        xs = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
        dys = [torch.randn(N, H, H, Cout, device="cuda").to(dt) for _ in range(nset)]
        w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05).to(dt)

        def call(i):
            if kind == "fwd":
                return ops.conv2d_fwd(xs[i], w, 1, 0)
            if kind == "dgrad":
                return ops.conv2d_dgrad(dys[i], w, (N, H, H, Cin), 1, 0)
            return ops.conv2d_dgrad(dys[i], w, (N, H, H, Cin), 1, 0, addend=xs[i])

        res = []
        ncols = Cout if kind == "fwd" else Cin
        for v in VARIANTS:
            if v not in (None, "0") and ncols % int(v.rstrip("f").split("x")[1]):
                continue
            if v is None:
                os.environ.pop("MI355_IGEMM8", None)
                native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
            else:
                os.environ["MI355_IGEMM8"] = v
                native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
            for i in range(nset):
                call(i)
            torch.cuda.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                for i in range(nset):
                    call(i)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / (reps * nset)
            res.append(f"{v or 'rule':8s} {t * 1e6:6.1f}us {by / t / 1e12:4.2f}TB/s")
        os.environ["MI355_IGEMM8"] = "0"
        native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
        for big in ("1", "3"):  # the 4-wave 256x256 tile / the 8-wave 3-stage 256x128 tile of conv_igemm.hip, forced
            if big == "1" and ncols % 256:
                continue
            os.environ["MI355_IGEMM_BIG"] = big
            native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
            for i in range(nset):
                call(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                for i in range(nset):
                    call(i)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / (3 * nset)
            res.append(f"big{big} {t * 1e6:6.1f}us {by / t / 1e12:4.2f}TB/s")
        os.environ.pop("MI355_IGEMM_BIG", None)
        native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
        os.environ.pop("MI355_IGEMM8", None)
        native.lib().mi355_reload_knobs()  # (the library reads its tile knobs once)
        # streaming yardstick with the same bytes: out = a + b style kernels over bf16 buffers
        n_el = by // 2 // 3
        a = [torch.empty(n_el, device="cuda", dtype=dt) for _ in range(nset)]
        b = [torch.empty(n_el, device="cuda", dtype=dt) for _ in range(nset)]
        c = [torch.empty(n_el, device="cuda", dtype=dt) for _ in range(nset)]
        for i in range(nset):
            torch.add(a[i], b[i], out=c[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for i in range(nset):
                torch.add(a[i], b[i], out=c[i])
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / (3 * nset)
        res.append(f"torch.add {t * 1e6:6.1f}us {n_el * 6 / t / 1e12:4.2f}TB/s")
        print(f"{name:16s} {by / 1e6:6.0f}MB | " + " | ".join(res), flush=True)
        del xs, dys, a, b, c
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
