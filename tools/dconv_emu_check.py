"""dconv_emu_check.py — run a generated direct-conv kernel (csrc/asm/dconv_gen.py) in the CPU emulator (tools/gcn_emu.py)
against a numpy convolution on exact small-integer data.  Test infrastructure; used by tests/test_dconv_emu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))

import dconv_gen  # noqa: E402
import gcn_emu  # noqa: E402


def to_bf16_bits(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    assert ((u & 0xFFFF) == 0).all(), "test data must be exactly representable in bf16"
    return (u >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def bf16_round(x):
    return from_bf16_bits(gcn_emu.bf16_round_rne(np.asarray(x, dtype=np.float32)).astype(np.uint16))


def to_e4m3_bytes(x):
    """exactly representable values -> OCP e4m3fn bytes (the decode table of the emulator, inverted)"""
    table = gcn_emu.Emulator._e4m3(np.arange(256))
    lut = {float(v): i for i, v in enumerate(table) if not np.isnan(v) and not (i == 0x80)}
    flat = np.asarray(x, dtype=np.float64).ravel()
    out = np.array([lut[float(v)] for v in flat], dtype=np.uint8)
    return out.reshape(np.shape(x))


def conv_ref(x, w, taps):
    """x [N][H][W][C], w [Co][9][C] (float64); taps: list of 9 (dh, dw, wtap) -> out [N][H][W][Co]"""
    N, H, W, C = x.shape
    Co = w.shape[0]
    xp = np.zeros((N, H + 2, W + 2, C))
    xp[:, 1:H + 1, 1:W + 1] = x
    out = np.zeros((N, H, W, Co))
    for dh, dw, wt in taps:
        out += np.einsum("nhwc,oc->nhwo", xp[:, 1 + dh:1 + dh + H, 1 + dw:1 + dw + W], w[:, wt])
    return out


def run(name, tiles=(0,), ntile=0, seed=0, dgrad_taps=False, check=True, over_scales=True, **over):
    c, g, text = dconv_gen.generate(name, **over)
    rng = np.random.default_rng(seed)
    ntiles = max(tiles) + 1
    N = -(-ntiles // c.TPI) if c.ROWS_T else ntiles * c.IPT
    x = rng.integers(-2, 3, size=(N, c.H, c.W, c.Cin)).astype(np.float32)
    w = rng.integers(-2, 3, size=(c.NCOLS, 9, c.Cin)).astype(np.float32)
    if dgrad_taps:
        taps = [(1 - kh, 1 - kw, kh * 3 + kw) for kh in range(3) for kw in range(3)]
    else:
        taps = [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)]
    # the kernel's tap t = (dh + 1)*3 + (dw + 1) reads weight tap wtap
    wtap_of = [0] * 9
    for dh, dw, wt in taps:
        wtap_of[(dh + 1) * 3 + (dw + 1)] = wt
    mem = gcn_emu.Memory()
    fp8 = getattr(c, "fp8", 0)
    a_in = mem.alloc(to_e4m3_bytes(x) if fp8 else to_bf16_bits(x))
    a_wt = mem.alloc(to_e4m3_bytes(w) if fp8 else to_bf16_bits(w))
    out0 = np.full((N, c.H, c.W, c.NCOLS), 0x7FC0, dtype=np.uint16)  # NaN: unwritten outputs show
    a_out = mem.alloc(out0)
    ntiles_all = N * c.TPI if c.ROWS_T else ntiles
    a_stat = mem.alloc(np.full((ntiles_all, 2, c.NCOLS), np.nan, dtype=np.float32))
    # BN-backward inputs (stats == 2): y laid out like the output, one mask byte per 8 channels, per-channel mean / invstd
    yb = rng.integers(-3, 4, size=(N, c.H, c.W, c.NCOLS)).astype(np.float32)
    bits = rng.integers(0, 256, size=(N, c.H, c.W, c.NCOLS // 8)).astype(np.uint8)
    mean = (rng.integers(-4, 5, size=c.NCOLS) * 0.25).astype(np.float32)
    invstd = (rng.integers(1, 5, size=c.NCOLS) * 0.5).astype(np.float32)
    a_y, a_bits, a_mu, a_is = mem.alloc(to_bf16_bits(yb)), mem.alloc(bits), mem.alloc(mean), mem.alloc(invstd)
    if fp8:
        # e4m3 operands: out = acc * oscale / (scale_in * scale_wt); powers of two keep the integer test exact.  scales=False: oscale alone (null pointers)
        oscale, sc_in, sc_wt = (0.5, 2.0, 0.25) if over_scales else (0.5, None, None)
        a_si = mem.alloc(np.array([sc_in or 1.0], dtype=np.float32))
        a_sw = mem.alloc(np.array([sc_wt or 1.0], dtype=np.float32))
        osc_eff = oscale / ((sc_in or 1.0) * (sc_wt or 1.0))
        fields = [("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat), ("q", a_y), ("q", a_bits), ("q", a_mu), ("q", a_is),
                  ("q", a_si if sc_in else 0)] + [("I", wtap_of[t] * c.Cin) for t in range(9)] + [("I", c.Cin // 128)] + [("I", 0), ("f", oscale), ("q", a_sw if sc_in else 0)]
    else:
        osc_eff = 1.0
        fields = [("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat), ("q", a_y), ("q", a_bits), ("q", a_mu), ("q", a_is),
                  ("q", 0)] + [("I", wtap_of[t] * c.Cin * 2) for t in range(9)] + [("I", c.Cin // 64)] + [("I", 0)] * 4
    fields += [("I", w) for par in dconv_gen.tables(c) for row in par for w in row]
    ka = gcn_emu.pack_kernarg(fields)
    assert len(ka) == dconv_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for t in tiles:
        emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check, dontcare=[(c.ABASE + (b + 1) * c.ABUF, c.ABASE + (b + 1) * c.ABUF + 256) for b in range(2)])
        total += emu.run_workgroup(4, a_ka, wg_id=(t, ntile, 0))
    got = from_bf16_bits(mem.array(a_out, np.uint16, out0.shape)).astype(np.float64)
    ref = conv_ref(x.astype(np.float64), w.astype(np.float64), taps) * osc_eff
    cols = slice(ntile * c.BN, ntile * c.BN + c.BN)
    res = {"insts": total, "cfg": c}
    refr = bf16_round(ref.astype(np.float32)).astype(np.float64)
    # tiles as flat pixel-row ranges: tile t covers rows [t*tile_rows, (t+1)*tile_rows) of the [N*H][W] pixel grid
    got = got.reshape(N * c.H, c.W, c.NCOLS)
    refr = refr.reshape(N * c.H, c.W, c.NCOLS)
    rows_of = lambda t: slice(t * c.tile_rows, (t + 1) * c.tile_rows)
    res["max_err"] = float(max(np.abs(got[rows_of(t)][..., cols] - refr[rows_of(t)][..., cols]).max() for t in tiles))
    res["untouched_ok"] = True
    if c.NCOLS > c.BN:
        other = np.ones(c.NCOLS, dtype=bool)
        other[cols] = False
        res["untouched_ok"] = bool(np.isnan(got[..., other]).all())
    if c.stats == 1:
        st = mem.array(a_stat, np.float32, (ntiles_all, 2, c.NCOLS))
        s1 = np.stack([refr[rows_of(t)].sum(axis=(0, 1)) for t in range(ntiles_all)])
        s2 = np.stack([(refr[rows_of(t)] ** 2).sum(axis=(0, 1)) for t in range(ntiles_all)])
        tl = list(tiles)
        res["stat_err"] = float(max(np.abs(st[tl, 0][:, cols] - s1[tl][:, cols]).max(), np.abs(st[tl, 1][:, cols] - s2[tl][:, cols]).max()))
    if c.stats >= 2:
        st = mem.array(a_stat, np.float32, (ntiles_all, 2, c.NCOLS))
        mask = ((bits[..., None] >> np.arange(8)) & 1).reshape(N * c.H, c.W, c.NCOLS).astype(np.float64)
        # stats 3: the leaky-ReLU mask — the fp32 product of the stored (bf16) value and 0.01f where the bit is clear
        dz = refr * mask if c.stats == 2 else np.where(mask > 0, refr, (refr.astype(np.float32) * np.float32(0.01)).astype(np.float64))
        xhat = (yb.astype(np.float64).reshape(N * c.H, c.W, c.NCOLS) - mean) * invstd
        tl = list(tiles)
        s1 = np.stack([dz[rows_of(t)].sum(axis=(0, 1)) for t in range(ntiles_all)])
        s2 = np.stack([(dz * xhat)[rows_of(t)].sum(axis=(0, 1)) for t in range(ntiles_all)])
        scale = max(np.abs(s1).max(), np.abs(s2).max(), 1.0)
        res["stat_err"] = float(max(np.abs(st[tl, 0][:, cols] - s1[tl][:, cols]).max(), np.abs(st[tl, 1][:, cols] - s2[tl][:, cols]).max()) / scale)
    return res


def run_s2d(name, tiles=(0,), ntile=0, classes=(0, 1, 2, 3), seed=0, check=True, **over):
    """a stride-2 data-gradient kernel (Cfg.s2d): dx[n][2i + ph][2j + pw] = sum over the class's taps of dy[n][i + dh][j + dw] * w[.][kh*3 + kw][.] with
    kh = ph + 1 - 2 dh, kw = pw + 1 - 2 dw (pad 1); one workgroup per (tile, class), checked against the transposed convolution in numpy"""
    c, g, text = dconv_gen.generate(name, **over)
    assert c.s2d
    rng = np.random.default_rng(seed)
    ntiles = max(tiles) + 1
    N = -(-ntiles // c.TPI) if c.ROWS_T else ntiles * c.IPT
    H, W, H2, W2 = c.H, c.W, 2 * c.H, 2 * c.W
    dy = rng.integers(-2, 3, size=(N, H, W, c.Cin)).astype(np.float32)
    w = rng.integers(-2, 3, size=(c.NCOLS, 9, c.Cin)).astype(np.float32)      # [Ncols][wtaps][Ck]: the transposed weights of the data gradient
    wtap_of, ref = [0] * 9, np.zeros((N, H2, W2, c.NCOLS))
    dyp = np.zeros((N, H + 1, W + 1, c.Cin))
    dyp[:, :H, :W] = dy
    slot = 0
    for (ph, pw), taps in dconv_gen.Gen.S2D_CLASSES:
        for dh, dw in taps:
            kh, kw = ph + 1 - 2 * dh, pw + 1 - 2 * dw
            assert 0 <= kh < 3 and 0 <= kw < 3
            wtap_of[slot] = kh * 3 + kw
            slot += 1
            ref[:, ph::2, pw::2] += np.einsum("nhwc,oc->nhwo", dyp[:, dh:dh + H, dw:dw + W], w[:, kh * 3 + kw].astype(np.float64))
    mem = gcn_emu.Memory()
    a_in, a_wt = mem.alloc(to_bf16_bits(dy)), mem.alloc(to_bf16_bits(w))
    out0 = np.full((N, H2, W2, c.NCOLS), 0x7FC0, dtype=np.uint16)
    a_out = mem.alloc(out0)
    ntiles_all = N * c.TPI if c.ROWS_T else ntiles
    a_stat = mem.alloc(np.full((ntiles_all * 4, 2, c.NCOLS), np.nan, dtype=np.float32))
    yb = rng.integers(-3, 4, size=(N, H2, W2, c.NCOLS)).astype(np.float32)
    bits = rng.integers(0, 256, size=(N, H2, W2, c.NCOLS // 8)).astype(np.uint8)
    mean = (rng.integers(-4, 5, size=c.NCOLS) * 0.25).astype(np.float32)
    invstd = (rng.integers(1, 5, size=c.NCOLS) * 0.5).astype(np.float32)
    a_y, a_bits, a_mu, a_is = mem.alloc(to_bf16_bits(yb)), mem.alloc(bits), mem.alloc(mean), mem.alloc(invstd)
    fields = [("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat), ("q", a_y), ("q", a_bits), ("q", a_mu), ("q", a_is),
              ("q", 0)] + [("I", wtap_of[t] * c.Cin * 2) for t in range(9)] + [("I", c.Cin // 64)] + [("I", 0)] * 4
    fields += [("I", x) for par in dconv_gen.tables(c) for row in par for x in row]
    ka = gcn_emu.pack_kernarg(fields)
    assert len(ka) == dconv_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    nct = c.NCOLS // c.BN
    total = 0
    for t in tiles:
        for k in classes:
            emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check, dontcare=[(c.ABASE + (b + 1) * c.ABUF, c.ABASE + (b + 1) * c.ABUF + 256) for b in range(2)])
            total += emu.run_workgroup(4, a_ka, wg_id=(t, k * nct + ntile, 0))
    got = from_bf16_bits(mem.array(a_out, np.uint16, out0.shape)).astype(np.float64).reshape(N * H2, W2, c.NCOLS)
    refr = bf16_round(ref.astype(np.float32)).astype(np.float64).reshape(N * H2, W2, c.NCOLS)
    cols = slice(ntile * c.BN, ntile * c.BN + c.BN)
    rows_of = lambda t: slice(t * 2 * c.tile_rows, (t + 1) * 2 * c.tile_rows)
    res = {"insts": total, "cfg": c, "max_err": 0.0, "untouched_ok": True, "stat_err": 0.0}
    mask = ((bits[..., None] >> np.arange(8)) & 1).reshape(N * H2, W2, c.NCOLS).astype(np.float64)
    dz = refr * mask
    dzx = dz * ((yb.astype(np.float64).reshape(N * H2, W2, c.NCOLS) - mean) * invstd)
    st = mem.array(a_stat, np.float32, (ntiles_all * 4, 2, c.NCOLS))
    scale = max(np.abs(dz).sum(axis=(0, 1)).max(), np.abs(dzx).sum(axis=(0, 1)).max(), 1.0)
    for t in tiles:
        for k, ((ph, pw), _) in enumerate(dconv_gen.Gen.S2D_CLASSES):
            g_, r_ = got[rows_of(t)][ph::2, pw::2][..., cols], refr[rows_of(t)][ph::2, pw::2][..., cols]
            if k in classes:
                res["max_err"] = max(res["max_err"], float(np.abs(g_ - r_).max()))   # (NaN = an unwritten output: fails the == 0.0 assertion)
                if np.isnan(g_).any():
                    res["max_err"] = float("nan")
                if c.stats == 2:
                    s1 = dz[rows_of(t)][ph::2, pw::2].sum(axis=(0, 1))[cols]
                    s2 = dzx[rows_of(t)][ph::2, pw::2].sum(axis=(0, 1))[cols]
                    res["stat_err"] = max(res["stat_err"], float(max(np.abs(st[4 * t + k, 0][cols] - s1).max(), np.abs(st[4 * t + k, 1][cols] - s2).max()) / scale))
            else:
                res["untouched_ok"] &= bool(np.isnan(g_).all())
    if c.NCOLS > c.BN:
        other = np.ones(c.NCOLS, dtype=bool)
        other[cols] = False
        res["untouched_ok"] &= bool(np.isnan(got[..., other]).all())
    return res


def run_bn(name, tiles=(0,), ntile=0, seed=0, check=True, **over):
    """a forward kernel with the previous layer's BatchNorm + ReLU in its operand path (Cfg.bnin): out = conv(a), a = bf16(relu(y * scale + shift)) — what
    bn_apply_kernel computes —, with a and its ReLU bits (1 byte per 8 channels) left in memory by the kernel; dyadic data makes every step exact"""
    c, g, text = dconv_gen.generate(name, **over)
    assert c.bnin
    rng = np.random.default_rng(seed)
    ntiles = max(tiles) + 1
    N = -(-ntiles // c.TPI) if c.ROWS_T else ntiles * c.IPT
    y = rng.integers(-3, 4, size=(N, c.H, c.W, c.Cin)).astype(np.float32)
    scale = (rng.integers(1, 9, size=c.Cin) * 0.25).astype(np.float32)
    shift = (rng.integers(-8, 9, size=c.Cin) * 0.25).astype(np.float32)
    w = rng.integers(-2, 3, size=(c.NCOLS, 9, c.Cin)).astype(np.float32)
    v = y * scale + shift                       # exact: multiples of 1/16 below 16
    a = np.maximum(v, 0).astype(np.float32)
    assert (bf16_round(a) == a).all()
    abits = np.packbits((v > 0).reshape(N, c.H, c.W, c.Cin // 8, 8), axis=-1, bitorder="little")[..., 0]
    taps = [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)]
    mem = gcn_emu.Memory()
    a_in, a_wt = mem.alloc(to_bf16_bits(y)), mem.alloc(to_bf16_bits(w))
    out0 = np.full((N, c.H, c.W, c.NCOLS), 0x7FC0, dtype=np.uint16)
    a_out = mem.alloc(out0)
    ntiles_all = N * c.TPI if c.ROWS_T else ntiles
    a_stat = mem.alloc(np.full((ntiles_all, 2, c.NCOLS), np.nan, dtype=np.float32))
    a_a = mem.alloc(np.full((N, c.H, c.W, c.Cin), 0x7FC0, dtype=np.uint16))
    a_bits = mem.alloc(np.full((N, c.H, c.W, c.Cin // 8), 0x55, dtype=np.uint8))
    a_ss = mem.alloc(np.concatenate([scale, shift]))
    fields = [("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat), ("q", a_a), ("q", a_bits), ("q", a_ss), ("q", 0),
              ("q", 0)] + [("I", t * c.Cin * 2) for t in range(9)] + [("I", c.Cin // 64)] + [("I", 0)] * 4
    fields += [("I", x) for par in dconv_gen.tables(c) for row in par for x in row]
    fields += [("I", x) for par in dconv_gen.ttables(c) for row in par for x in row]
    ka = gcn_emu.pack_kernarg(fields)
    assert len(ka) == g.ka_size, (len(ka), g.ka_size)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for t in tiles:
        emu = gcn_emu.Emulator(text, mem, lds_bytes=g.lds_bytes, check=check, dontcare=[(c.ABASE + (b + 1) * c.ABUF, c.ABASE + (b + 1) * c.ABUF + 256) for b in range(2)])
        total += emu.run_workgroup(4, a_ka, wg_id=(t, ntile, 0))
    got = from_bf16_bits(mem.array(a_out, np.uint16, out0.shape)).astype(np.float64).reshape(N * c.H, c.W, c.NCOLS)
    refr = bf16_round(conv_ref(a.astype(np.float64), w.astype(np.float64), taps).astype(np.float32)).astype(np.float64).reshape(N * c.H, c.W, c.NCOLS)
    cols = slice(ntile * c.BN, ntile * c.BN + c.BN)
    rows_of = lambda t: slice(t * c.tile_rows, (t + 1) * c.tile_rows)
    res = {"insts": total, "cfg": c}
    res["max_err"] = float(max(np.abs(got[rows_of(t)][..., cols] - refr[rows_of(t)][..., cols]).max() for t in tiles))
    if c.stats == 1:
        st = mem.array(a_stat, np.float32, (ntiles_all, 2, c.NCOLS))
        tl = list(tiles)
        s1 = np.stack([refr[rows_of(t)].sum(axis=(0, 1)) for t in tl])
        s2 = np.stack([(refr[rows_of(t)] ** 2).sum(axis=(0, 1)) for t in tl])
        # (relative: the outputs are multiples of 1/16 here, their squares do not add exactly in fp32)
        res["stat_err"] = float(max(np.abs(st[tl, 0][:, cols] - s1[:, cols]).max() / max(1.0, np.abs(s1).max()), np.abs(st[tl, 1][:, cols] - s2[:, cols]).max() / max(1.0, np.abs(s2).max())))
    # the by-products: every pixel of the tiles' own rows written and right; whatever else was written (halo rows of neighbouring tiles) right too
    ga = from_bf16_bits(mem.array(a_a, np.uint16, (N, c.H, c.W, c.Cin))).reshape(N * c.H, c.W, c.Cin)
    gb = mem.array(a_bits, np.uint8, (N, c.H, c.W, c.Cin // 8)).reshape(N * c.H, c.W, c.Cin // 8)
    ra, rb = a.reshape(N * c.H, c.W, c.Cin), abits.reshape(N * c.H, c.W, c.Cin // 8)
    own = np.zeros(N * c.H, dtype=bool)
    for t in tiles:
        own[rows_of(t)] = True
    res["a_ok"] = bool((ga[own] == ra[own]).all() and (np.isnan(ga[~own]) | (ga[~own] == ra[~own])).all())
    res["bits_ok"] = bool((gb[own] == rb[own]).all() and ((gb[~own] == 0x55) | (gb[~own] == rb[~own])).all())
    return res


if __name__ == "__main__":
    import time
    t0 = time.time()
    r = run("dconv_l3_s1", tiles=(0, 1), Cin=128)
    print({k: v for k, v in r.items() if k != "cfg"}, "%.1f s" % (time.time() - t0))
