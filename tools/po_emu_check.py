"""po_emu_check.py — run a generated output-heavy pointwise kernel (csrc/asm/po_gen.py) in the CPU emulator (tools/gcn_emu.py) against numpy
on exact small-integer data: output (with the shortcut addend under its ReLU bits), BN statistics rows / BN-backward sums, untouched memory.
Test infrastructure; used by tests/test_dconv_emu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))

import gcn_emu  # noqa: E402
import po_gen  # noqa: E402
from dconv_emu_check import bf16_round, from_bf16_bits, to_bf16_bits  # noqa: E402


def plan(M, N, BN, TP, cus=256):
    """the host-side launch plan of dconv.cpp launch_po(): tiles, column tiles, tiles per group, groups, grid"""
    T = -(-M // TP)
    nct = N // BN
    gmax = max(1, cus // nct)
    tpg = -(-T // gmax)
    G = -(-T // tpg)
    grid = -(-G // 8) * 8 * nct
    return T, nct, tpg, G, grid


def wg_of(g, ct, nct):
    """workgroup id of (group, column tile): xcd = g % 8, l = (g // 8) * nct + ct"""
    return ((g // 8) * nct + ct) * 8 + g % 8


def run(name, M=200, N=None, groups=((0, 0),), tpg=None, seed=0, check=True, extra_wgs=(), HW=(10, 20), **over):
    c, g, text = po_gen.generate(name, **over)
    N = N or c.BN
    rng = np.random.default_rng(seed)
    T = -(-M // c.TP)
    nct = N // c.BN
    tpg = tpg or T
    G = -(-T // tpg)
    x = rng.integers(-2, 3, size=(M, c.K)).astype(np.float32)
    w = rng.integers(-2, 3, size=(N, c.K)).astype(np.float32)
    bn = getattr(c, "bnin", 0)
    if bn:   # the input is raw: a = relu(x * scale + shift) (dyadic: exact), the kernel multiplies a and leaves a + its ReLU bits in memory
        assert M % c.TP == 0
        x = rng.integers(-3, 4, size=(M, c.K)).astype(np.float32)
        scale = (rng.integers(1, 9, size=c.K) * 0.25).astype(np.float32)
        shift = (rng.integers(-8, 9, size=c.K) * 0.25).astype(np.float32)
        v_ = x * scale + shift
        a_act = np.maximum(v_, 0).astype(np.float32)
        assert (bf16_round(a_act) == a_act).all()
        act_bits = np.packbits((v_ > 0).reshape(M, c.K // 8, 8), axis=-1, bitorder="little")[..., 0]
    ad = rng.integers(-3, 4, size=(M, N)).astype(np.float32)
    H, W = HW
    if c.add == 3:
        # the addend at half resolution: M = images * H * W pixels; only pixels with even row and column have one
        assert M % (H * W) == 0 and H % 2 == 0 and W % 2 == 0
        nimg = M // (H * W)
        adc = rng.integers(-3, 4, size=(nimg, H // 2, W // 2, N)).astype(np.float32)
        full = np.zeros((nimg, H, W, N), dtype=np.float32)
        full[:, ::2, ::2] = adc
        ad = full.reshape(M, N)
    abits = rng.integers(0, 256, size=(M, N // 8)).astype(np.uint8)
    yb = rng.integers(-3, 4, size=(M, N)).astype(np.float32)
    bits = rng.integers(0, 256, size=(M, N // 8)).astype(np.uint8)
    mean = (rng.integers(-4, 5, size=N) * 0.25).astype(np.float32)
    invstd = (rng.integers(1, 5, size=N) * 0.5).astype(np.float32)
    mem = gcn_emu.Memory()
    a_in, a_wt = mem.alloc(to_bf16_bits(x)), mem.alloc(to_bf16_bits(w))
    out0 = np.full((M, N), 0x7FC0, dtype=np.uint16)
    a_out = mem.alloc(out0)
    a_stat = mem.alloc(np.full((G * c.WM, 2, N), np.nan, dtype=np.float32))   # (waves 2 x 2: the two waves that share columns leave a row each)
    a_y, a_bits, a_mu, a_is = mem.alloc(to_bf16_bits(yb)), mem.alloc(bits), mem.alloc(mean), mem.alloc(invstd)
    if bn:   # the pointer slots of the BN-backward sums carry the input's BatchNorm: a out, its bits out, [2][K] scale / shift
        a_y = mem.alloc(np.full((M, c.K), 0x7FC0, dtype=np.uint16))
        a_bits = mem.alloc(np.full((M, c.K // 8), 0x55, dtype=np.uint8))
        a_mu = mem.alloc(np.concatenate([scale, shift]))
    a_ad, a_ab = mem.alloc(to_bf16_bits(adc if c.add == 3 else ad)), mem.alloc(abits)
    lognct = nct.bit_length() - 1
    assert 1 << lognct == nct
    ka = gcn_emu.pack_kernarg([("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat), ("q", a_y), ("q", a_bits), ("q", a_mu), ("q", a_is), ("q", a_ad), ("q", a_ab),
                               ("I", M), ("I", N), ("I", tpg), ("I", G), ("I", T), ("I", lognct), ("I", W), ("I", H), ("I", (1 << 32) // W + 1), ("I", (1 << 32) // H + 1),
                               ("I", 0), ("I", 0)])
    assert len(ka) == po_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for (gi, ct) in groups:
        emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check)
        total += emu.run_workgroup(4, a_ka, wg_id=(wg_of(gi, ct, nct), 0, 0))
    for gi in extra_wgs:  # run indices >= G: the workgroup must end without a store
        emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check)
        assert gi >= G
        total += emu.run_workgroup(4, a_ka, wg_id=(wg_of(gi, 0, nct), 0, 0))
    got = from_bf16_bits(mem.array(a_out, np.uint16, out0.shape)).astype(np.float64)
    ref = ((a_act if bn else x).astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)
    if c.add:
        amask = ((abits[..., None] >> np.arange(8)) & 1).reshape(M, N).astype(np.float32) if c.add == 2 else np.ones((M, N), dtype=np.float32)  # (add == 3: `ad` is the zero-filled full tensor)
        ref = ref + ad * amask
    refr = bf16_round(ref).astype(np.float64)
    res = {"insts": total, "cfg": c}
    touched = np.zeros(out0.shape, dtype=bool)
    rows_of = lambda gi: slice(gi * tpg * c.TP, min((gi + 1) * tpg * c.TP, M))
    cols_of = lambda ct: slice(ct * c.BN, (ct + 1) * c.BN)
    for (gi, ct) in groups:
        touched[rows_of(gi), cols_of(ct)] = True
    res["max_err"] = (float(np.abs(np.where(touched, got - refr, 0.0)).max()) if not np.isnan(got[touched]).any() else float("nan")) if touched.any() else 0.0
    res["untouched_ok"] = bool(np.isnan(got[~touched]).all())
    if bn:
        ga = from_bf16_bits(mem.array(a_y, np.uint16, (M, c.K)))
        gb = mem.array(a_bits, np.uint8, (M, c.K // 8))
        wrote = np.zeros(M, dtype=bool)
        for (gi, ct) in groups:
            if ct == 0:
                wrote[rows_of(gi)] = True
        res["a_ok"] = bool((ga[wrote] == a_act[wrote]).all() and np.isnan(ga[~wrote]).all())
        res["bits_ok"] = bool((gb[wrote] == act_bits[wrote]).all() and (gb[~wrote] == 0x55).all())
    st = mem.array(a_stat, np.float32, (G * c.WM, 2, N))
    if c.stats:
        stsum = st.reshape(G, c.WM, 2, N).astype(np.float64).sum(axis=1)
        err = 0.0
        scale = 1.0
        for (gi, ct) in groups:
            blk = refr[rows_of(gi), cols_of(ct)]
            if c.stats == 1:
                s1, s2 = blk.sum(axis=0), (blk ** 2).sum(axis=0)
            else:
                mask = ((bits[..., None] >> np.arange(8)) & 1).reshape(M, N).astype(np.float64)[rows_of(gi), cols_of(ct)]
                dz = blk * mask if c.stats == 2 else np.where(mask > 0, blk, (blk.astype(np.float32) * np.float32(0.01)).astype(np.float64))   # stats 3: leaky mask
                xhat = (yb.astype(np.float64)[rows_of(gi), cols_of(ct)] - mean[cols_of(ct)]) * invstd[cols_of(ct)]
                s1, s2 = dz.sum(axis=0), (dz * xhat).sum(axis=0)
            scale = max(scale, np.abs(s1).max(), np.abs(s2).max())
            err = max(err, np.abs(stsum[gi, 0, cols_of(ct)] - s1).max(), np.abs(stsum[gi, 1, cols_of(ct)] - s2).max())
        res["stat_err"] = float(err / scale)
        tst = np.zeros((G * c.WM, 2, N), dtype=bool)
        for (gi, ct) in groups:
            tst[gi * c.WM:(gi + 1) * c.WM, :, cols_of(ct)] = True
        res["untouched_ok"] = res["untouched_ok"] and bool(np.isnan(st[~tst]).all()) and not np.isnan(st[tst]).any()
    return res


if __name__ == "__main__":
    import time
    for name, kw in (("po_k64_b256_s1_a0", dict(M=200, groups=((0, 0),))),
                     ("po_k64_b256_s2_a2", dict(M=300, N=512, tpg=3, groups=((1, 1), (0, 0)))),):
        t0 = time.time()
        r = run(name, **kw)
        print(name, kw, {k: v for k, v in r.items() if k != "cfg"}, "%.1f s" % (time.time() - t0))
