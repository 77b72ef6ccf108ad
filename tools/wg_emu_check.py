"""wg_emu_check.py — run a generated 3x3 weight-gradient kernel (csrc/asm/wg_gen.py) in the CPU emulator (tools/gcn_emu.py) against a
numpy weight gradient on exact small-integer data.  Test infrastructure; used by tests/test_dconv_emu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))

import gcn_emu  # noqa: E402
import wg_gen  # noqa: E402
from dconv_emu_check import to_bf16_bits  # noqa: E402


def wgrad_ref(x, dy):
    """x [N][H][W][C], dy [N][H][W][Co] -> dw [Co][9][C] (pad 1, stride 1)"""
    N, H, W, C = x.shape
    xp = np.zeros((N, H + 2, W + 2, C))
    xp[:, 1:H + 1, 1:W + 1] = x
    out = np.zeros((dy.shape[-1], 9, C))
    for ky in range(3):
        for kx in range(3):
            out[:, ky * 3 + kx] = np.einsum("nhwo,nhwc->oc", dy, xp[:, ky:ky + H, kx:kx + W])
    return out


def run(name, pairs=((0, 0),), splits=2, tps=2, seed=0, check=True, **over):
    """pairs: (ci tile, co tile) workgroups to run, for every split; tps tiles per split"""
    c, g, text = wg_gen.generate(name, **over)
    rng = np.random.default_rng(seed)
    ntiles = splits * tps
    tn, ti = c.TPI_NUM
    N = -(-ntiles * ti // tn)   # whole images; row tiles may stop inside the last one (the kernel is given `ntiles`)
    assert tn > 1 or N * tn == ntiles * ti
    x = rng.integers(-2, 3, size=(N, c.H, c.W, c.C)).astype(np.float32)
    dy = rng.integers(-2, 3, size=(N, c.H, c.W, c.CO)).astype(np.float32)
    mem = gcn_emu.Memory()
    a_dy, a_x = mem.alloc(to_bf16_bits(dy)), mem.alloc(to_bf16_bits(x))
    p0 = np.full((splits, c.CO, 9, c.C), np.nan, dtype=np.float32)
    a_p = mem.alloc(p0)
    ka = gcn_emu.pack_kernarg([("q", a_dy), ("q", a_x), ("q", a_p), ("I", tps), ("I", ntiles)] + [("I", 0)] * 8)
    assert len(ka) == wg_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for ci_t, co_t in pairs:
        for sp in range(splits):
            emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check, dontcare=[((b + 1) * c.BUF, (b + 1) * c.BUF + 256) for b in range(2)])  # the two spill positions meet dy's zero columns
            total += emu.run_workgroup(4, a_ka, wg_id=(sp, ci_t + c.NCI * co_t, 0))
    got = mem.array(a_p, np.float32, p0.shape).astype(np.float64)
    # tiles of split s = images [s*tps*ti/tn, ...) (whole images only when tn == 1; row tiles: the split must cover whole images or the
    # reference is computed per row range)
    res = {"insts": total, "cfg": c, "max_err": 0.0, "untouched_ok": True}
    touched = np.zeros(p0.shape, dtype=bool)
    for sp in range(splits):
        if tn == 1:
            n0, n1 = sp * tps * ti, (sp + 1) * tps * ti
            ref = wgrad_ref(x[n0:n1].astype(np.float64), dy[n0:n1].astype(np.float64))
        else:
            # row tiles: dy rows of this split's tiles only, x whole (the halo rows come from neighbouring tiles)
            ref = np.zeros((c.CO, 9, c.C))
            for T in range(sp * tps, (sp + 1) * tps):
                n, t = divmod(T, tn)
                d = np.zeros_like(dy[n:n + 1], dtype=np.float64)
                d[:, t * c.DR:(t + 1) * c.DR] = dy[n, t * c.DR:(t + 1) * c.DR]
                ref += wgrad_ref(x[n:n + 1].astype(np.float64), d)
        for ci_t, co_t in pairs:
            a = got[sp, co_t * 64:(co_t + 1) * 64, :, ci_t * 64:(ci_t + 1) * 64]
            b = ref[co_t * 64:(co_t + 1) * 64, :, ci_t * 64:(ci_t + 1) * 64]
            res["max_err"] = max(res["max_err"], float(np.abs(a - b).max()) if not np.isnan(a).any() else float("inf"))
            touched[sp, co_t * 64:(co_t + 1) * 64, :, ci_t * 64:(ci_t + 1) * 64] = True
    res["untouched_ok"] = bool(np.isnan(got[~touched]).all())
    return res


if __name__ == "__main__":
    import time
    for name, kw in (("wg3_l3", dict(splits=2, tps=2)), ("wg3_l2", dict(splits=2, tps=4)), ("wg3_l4", dict(splits=2, tps=1))):
        t0 = time.time()
        r = run(name, pairs=((1, 0),), **kw)
        print(name, {k: v for k, v in r.items() if k != "cfg"}, "%.1f s" % (time.time() - t0))
