"""wg1_emu_check.py — run a generated 1x1 weight-gradient kernel (csrc/asm/wg1_gen.py) in the CPU emulator (tools/gcn_emu.py) against
numpy on exact small-integer data.  Test infrastructure; used by tests/test_dconv_emu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))

import gcn_emu  # noqa: E402
import wg1_gen  # noqa: E402
from dconv_emu_check import to_bf16_bits  # noqa: E402


def run(name, pairs=((0, 0),), splits=2, tps=2, npix=None, seed=0, check=True, **over):
    """pairs: (ci tile, co tile) workgroups to run, for every split; tps tiles of 64 pixels per split; npix: pixels of the tensors
    (default: exactly the tiles; fewer: the last tile is ragged)"""
    c, g, text = wg1_gen.generate(name, **over)
    rng = np.random.default_rng(seed)
    if npix is None:
        npix = splits * tps * c.TP
    ntiles = -(-npix // c.TP)
    x = rng.integers(-2, 3, size=(npix, c.C)).astype(np.float32)
    dy = rng.integers(-2, 3, size=(npix, c.CO)).astype(np.float32)
    mem = gcn_emu.Memory()
    a_dy, a_x = mem.alloc(to_bf16_bits(dy)), mem.alloc(to_bf16_bits(x))
    p0 = np.full((splits, c.CO, c.C), np.nan, dtype=np.float32)
    a_p = mem.alloc(p0)
    ka = gcn_emu.pack_kernarg([("q", a_dy), ("q", a_x), ("q", a_p), ("I", tps), ("I", ntiles), ("I", npix)] + [("I", 0)] * 7)
    assert len(ka) == wg1_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for ci_t, co_t in pairs:
        for sp in range(splits):
            emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check)
            total += emu.run_workgroup(4, a_ka, wg_id=(sp, ci_t + c.NCI * co_t, 0))
    got = mem.array(a_p, np.float32, p0.shape).astype(np.float64)
    res = {"insts": total, "cfg": c, "max_err": 0.0}
    touched = np.zeros(p0.shape, dtype=bool)
    for sp in range(splits):
        p_lo, p_hi = sp * tps * c.TP, min((sp + 1) * tps * c.TP, npix)
        ref = dy[p_lo:p_hi].astype(np.float64).T @ x[p_lo:p_hi].astype(np.float64)
        for ci_t, co_t in pairs:
            sl = (sp, slice(co_t * 64 * c.DP, (co_t + 1) * 64 * c.DP), slice(ci_t * 64 * c.XP, (ci_t + 1) * 64 * c.XP))
            a = got[sl]
            res["max_err"] = max(res["max_err"], float(np.abs(a - ref[sl[1], sl[2]]).max()) if not np.isnan(a).any() else float("inf"))
            touched[sl] = True
    res["untouched_ok"] = bool(np.isnan(got[~touched]).all())
    return res


if __name__ == "__main__":
    import time
    for name, kw in (("wg1_c256_o1024", dict(splits=2, tps=4, pairs=((1, 3),))), ("wg1_c1024_o256", dict(splits=1, tps=5, npix=300, pairs=((7, 0),))),
                     ("wg1_c512_o2048", dict(splits=2, tps=1, pairs=((0, 0),)))):
        t0 = time.time()
        r = run(name, **kw)
        print(name, kw, {k: v for k, v in r.items() if k != "cfg"}, "%.1f s" % (time.time() - t0))
