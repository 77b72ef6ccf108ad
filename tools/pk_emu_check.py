"""pk_emu_check.py — run a generated long-reduction pointwise kernel (csrc/asm/pk_gen.py) in the CPU emulator (tools/gcn_emu.py) against
numpy on exact small-integer data (output, BN statistics rows / BN-backward sums).  Test infrastructure; used by tests/test_dconv_emu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))

import gcn_emu  # noqa: E402
import pk_gen  # noqa: E402
from dconv_emu_check import bf16_round, from_bf16_bits, to_bf16_bits  # noqa: E402


def run(name, tiles=(0,), ntile=0, ntiles=None, seed=0, check=True, **over):
    c, g, text = pk_gen.generate(name, **over)
    rng = np.random.default_rng(seed)
    ntiles = ntiles or max(tiles) + 1
    M = ntiles * c.W
    x = rng.integers(-2, 3, size=(M, c.Cin)).astype(np.float32)
    w = rng.integers(-2, 3, size=(c.NCOLS, c.Cin)).astype(np.float32)
    mem = gcn_emu.Memory()
    a_in, a_wt = mem.alloc(to_bf16_bits(x)), mem.alloc(to_bf16_bits(w))
    out0 = np.full((M, c.NCOLS), 0x7FC0, dtype=np.uint16)
    a_out = mem.alloc(out0)
    a_stat = mem.alloc(np.full((ntiles, 2, c.NCOLS), np.nan, dtype=np.float32))
    yb = rng.integers(-3, 4, size=(M, c.NCOLS)).astype(np.float32)
    bits = rng.integers(0, 256, size=(M, c.NCOLS // 8)).astype(np.uint8)
    mean = (rng.integers(-4, 5, size=c.NCOLS) * 0.25).astype(np.float32)
    invstd = (rng.integers(1, 5, size=c.NCOLS) * 0.5).astype(np.float32)
    a_y, a_bits, a_mu, a_is = mem.alloc(to_bf16_bits(yb)), mem.alloc(bits), mem.alloc(mean), mem.alloc(invstd)
    ka = gcn_emu.pack_kernarg([("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat), ("q", a_y), ("q", a_bits), ("q", a_mu), ("q", a_is), ("q", 0),
                               ("I", c.Cin // 64)] + [("I", 0)] * 13)
    assert len(ka) == pk_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for t in tiles:
        emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check)
        total += emu.run_workgroup(4, a_ka, wg_id=(t, ntile, 0))
    got = from_bf16_bits(mem.array(a_out, np.uint16, out0.shape)).astype(np.float64)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    refr = bf16_round(ref.astype(np.float32)).astype(np.float64)
    cols = slice(ntile * c.BN, ntile * c.BN + c.BN)
    rows = lambda t: slice(t * c.W, (t + 1) * c.W)
    res = {"insts": total, "cfg": c}
    res["max_err"] = float(max(np.abs(got[rows(t), cols] - refr[rows(t), cols]).max() for t in tiles))
    touched = np.zeros(out0.shape, dtype=bool)
    for t in tiles:
        touched[rows(t), cols] = True
    res["untouched_ok"] = bool(np.isnan(got[~touched]).all())
    st = mem.array(a_stat, np.float32, (ntiles, 2, c.NCOLS))
    tl = list(tiles)
    if c.stats == 1:
        s1 = np.stack([refr[rows(t)].sum(axis=0) for t in range(ntiles)])
        s2 = np.stack([(refr[rows(t)] ** 2).sum(axis=0) for t in range(ntiles)])
        res["stat_err"] = float(max(np.abs(st[tl, 0][:, cols] - s1[tl][:, cols]).max(), np.abs(st[tl, 1][:, cols] - s2[tl][:, cols]).max()))
    if c.stats >= 2:
        mask = ((bits[..., None] >> np.arange(8)) & 1).reshape(M, c.NCOLS).astype(np.float64)
        dz = refr * mask if c.stats == 2 else np.where(mask > 0, refr, (refr.astype(np.float32) * np.float32(0.01)).astype(np.float64))   # stats 3: leaky mask
        xhat = (yb.astype(np.float64) - mean) * invstd
        s1 = np.stack([dz[rows(t)].sum(axis=0) for t in range(ntiles)])
        s2 = np.stack([(dz * xhat)[rows(t)].sum(axis=0) for t in range(ntiles)])
        scale = max(np.abs(s1).max(), np.abs(s2).max(), 1.0)
        res["stat_err"] = float(max(np.abs(st[tl, 0][:, cols] - s1[tl][:, cols]).max(), np.abs(st[tl, 1][:, cols] - s2[tl][:, cols]).max()) / scale)
    return res


if __name__ == "__main__":
    import time
    for name, kw in (("pk_k1024_n256_w196_s1", dict(tiles=(1,), Cin=256)), ("pk_k1024_n256_w196_s2", dict(tiles=(0, 2), Cin=448)), ("pk_k2048_n512_w98_s1", dict(tiles=(1,), ntile=1, Cin=320)),
                     ("pk_k2048_n512_w98_s2", dict(tiles=(0,), ntile=1, Cin=192)), ("pk_k1024_n256_w196_s0", dict(tiles=(0,), Cin=64))):
        t0 = time.time()
        r = run(name, **kw)
        print(name, kw, {k: v for k, v in r.items() if k != "cfg"}, "%.1f s" % (time.time() - t0))
