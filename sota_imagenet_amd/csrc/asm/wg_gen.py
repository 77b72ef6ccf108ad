#!/usr/bin/env python3
"""wg_gen.py — generator of the hand-scheduled gfx950 (MI355X) weight-gradient kernels of the 3x3 / stride-1 convolutions.

What the kernels replace: the cuDNN weight-gradient under `loss.backward()` of the reference (call form
/root/reference/sota_imagenet/callbacks.py:316-317) for the 3x3 convolutions of ResNet-50 — the same WgradArgs contract as
conv_wgrad.hip (fp32 partial slabs [split][Cout][9][Cin], summed in a fixed order by splitk_reduce), selected in launch_wgrad().

    dW[co][ky][kx][ci] = sum over (n, y, x) of dy[n][y][x][co] * in[n][y + ky - 1][x + kx - 1][ci]

Structure (one workgroup = 4 waves = one wave per SIMD; grid = splits x (ci tiles x co tiles): the workgroups of a split share an XCD):
  tile        64 input channels x 64 output channels x ALL 9 taps: 144 accumulator tiles of 16 x 16; wave w owns input channels
              16w .. 16w + 15 (36 tiles, 144 AGPRs).  v_mfma_f32_16x16x32_bf16 with the in fragment as src0 and the dy fragment as
              src1: a lane ends with 4 consecutive ci of one co (16-byte stores into the slab).
  reduction   over POSITIONS of a pixel tile: the dy tile is staged as [DR rows x P positions][128 B] with zero right padding, the
              zero-haloed in tile as [(DR + 2) x P positions][128 B]; position p of dy meets position p + ky*P + kx of in, so ONE
              staged pair serves the 9 taps (the implicit-GEMM form stages in once per tap).  32 positions per MFMA k-step; the
              k index is permuted (lane group g takes positions 4g .. 4g+3 and 16+4g .. 16+4g+3 of the step) identically for both
              operands, which makes the 16-byte chunk swizzle ((position >> 1) & 3) * 2 invariant under the +16 and +P row shifts:
              every read offset is an immediate, and ds_read_b64_tr_b16 is bank-conflict-free at every tap shift.
  operands    both are k-major in memory ([pixel][channel]), the MFMA wants 8 consecutive k per lane: ds_read_b64_tr_b16
              (hardware transpose), 2 per fragment; per k-step a wave reads 4 dy + 9 in fragments for 36 MFMAs.
  schedule    tiles double-buffered in LDS (2 x ~64 KiB); the LDS-DMA pieces of tile t + 2 are issued behind the barrier of the
              last k-step of tile t (the buffer tile t read is free then) and in the first k-step of tile t + 1; ONE s_barrier per
              tile; fragment reads of step s + 1 between the MFMAs of step s.
  epilogue    accumulators straight from AGPRs to the fp32 slab of this split.
"""
import argparse
import os
import sys
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dconv_gen import Alloc, R  # noqa: E402


@dataclass
class WCfg:
    name: str
    H: int
    W: int
    P: int        # LDS row pitch in positions (multiple of 8; P - W >= 2, or P == W + 1 with the shared halo column)
    C: int        # channels of the input tensor (reduction-side operand `x`; also Ck of the slab)
    CO: int       # channels of dy
    geom: str     # "img": a tile is one image | "rows": DR rows of one image | "pack": IPT images of 8 rows each (H = 7)
    DR: int       # dy rows of a tile
    IPT: int = 1  # images per tile (pack)
    rspan: int = 100  # tuning: reads / LDS-DMA of a k-step are issued within the first rspan % of its MFMAs
    noxwin: bool = False  # tuning: the 9-fragment form even where the x window applies

    @property
    def K(self):          # positions per tile
        return self.DR * self.P

    @property
    def KS(self):
        assert self.K % 32 == 0
        return self.K // 32

    @property
    def XR(self):
        return self.DR + 2

    @property
    def DYB(self):        # bytes of the dy tile
        return self.K * 128

    @property
    def XBYTES(self):
        return self.XR * self.P * 128

    @property
    def BUF(self):
        return self.DYB + self.XBYTES

    @property
    def LDS(self):        # two buffers + the two positions the last zero columns of dy read past the in tile
        return (2 * self.BUF + 256 + 1023) // 1024 * 1024

    @property
    def TPW(self):        # tiles per descriptor window (window = one image for "rows", else the tile itself)
        return self.H // self.DR if self.geom == "rows" else 1

    @property
    def WIN_PIX(self):    # pixels of a window
        return (self.IPT if self.geom == "pack" else 1) * self.H * self.W

    @property
    def TPI_NUM(self):    # tiles per image as a fraction (tiles, images)
        return (self.TPW, 1) if self.geom == "rows" else (1, self.IPT if self.geom == "pack" else 1)

    @property
    def NCI(self):
        return self.C // 64

    @property
    def NCO(self):
        return self.CO // 64


def sw(row):
    """16-byte chunk swizzle of LDS row `row` (even: a 16-channel fragment = an aligned chunk pair stays contiguous)"""
    return ((row >> 1) & 3) * 2


def rows_of(c, kind):
    """[(LDS row, source image row relative to the tile's first row | None = zero, dead = writes zeros via OOB lanes)]"""
    out = []
    if c.geom == "img":
        if kind == "d":
            out = [(d, d) for d in range(c.H)]
        else:
            out = [(g, g - 1) for g in range(1, c.H + 1)]
    elif c.geom == "rows":
        if kind == "d":
            out = [(d, d) for d in range(c.DR)]
        else:
            out = [(g, g - 1) for g in range(c.XR)]   # rows outside the image are out of the descriptor window: zeros
    else:
        for r in range(c.IPT * 8):
            i, y = divmod(r, 8)
            src = i * c.H + y if y < c.H else None
            out.append((r if kind == "d" else r + 1, src))
    return out


def pieces(c, kind):
    """per wave: list of (LDS byte offset inside the buffer, source byte constant, 8-position block of the row, dead).  The pieces of a
    tile are dealt to the four waves so that piece j of wave w is affine in (w, j) (piece_plan): flat round-robin where the blocks per
    row divide or are divided by 4, else rows x blocks as 2 x 2 or 4 x 1 (widths of 40: five or six blocks per row)."""
    ch = c.CO if kind == "d" else c.C
    rowb = c.W * ch * 2
    base = 0 if kind == "d" else c.DYB
    bpr = c.P // 8
    nblk0 = -(-(c.W + (0 if kind == "d" else 1)) // 8)     # blocks of a row that hold data (dy: columns 0 .. W-1, in: 0 .. W)
    rows = rows_of(c, kind)
    nrows = len(rows)

    def piece(lrow, src, xb):
        return (base + (lrow * c.P + xb * 8) * 128, (src if src is not None else 0) * rowb, xb, src is None)

    def affine(pw):
        n = len(pw[0])
        return all(len(x) == n for x in pw) and all(
            pw[w][j][0] - pw[w][0][0] == pw[0][j][0] - pw[0][0][0] and (pw[w][j][3] or pw[w][0][3] or pw[0][j][3] or pw[0][0][3] or
                                                                         pw[w][j][1] - pw[w][0][1] == pw[0][j][1] - pw[0][0][1])
            for w in range(4) for j in range(n))

    # flat dealing (a block of padding only is written with zeros: all its lanes are out of range)
    nblk = nblk0
    while (nrows * nblk) % 4 and nblk < bpr:
        nblk += 1
    allp = [piece(lrow, src, xb) for lrow, src in rows for xb in range(nblk)]
    if len(allp) % 4 == 0:
        pw = [allp[w::4] for w in range(4)]
        if affine(pw):
            return pw
    # rows x blocks dealing: wave (wr, wb) takes rows = wr (mod a), blocks = wb (mod b)
    for a, b in ((2, 2), (4, 1), (1, 4)):
        nblk = -(-nblk0 // b) * b
        if nrows % a or nblk > bpr:
            continue
        pw = []
        for w in range(4):
            wr, wb = w // b, w % b
            pw.append([piece(lrow, src, xb) for (lrow, src) in rows[wr::a] for xb in range(wb, nblk, b)])
        if affine(pw):
            return pw
    raise AssertionError("no affine dealing of the %s pieces: %d rows x %d blocks" % (kind, nrows, nblk0))


def piece_plan(c, kind):
    """affine form of the pieces: lds[w][j] = LW[w] + LJ[j], src[w][j] = SW[w] + SJ[j]; per j a VARIANT = (block of the row per wave,
    waves whose piece must write zeros): one lane-offset register per variant"""
    pw = pieces(c, kind)
    n = len(pw[0])
    LJ = [pw[0][j][0] - pw[0][0][0] for j in range(n)]
    SJ = [pw[0][j][1] - pw[0][0][1] for j in range(n)]
    LW = [pw[w][0][0] for w in range(4)]
    SW = [pw[w][0][1] for w in range(4)]
    keys = []
    for j in range(n):
        for w in range(4):
            assert pw[w][j][0] == LW[w] + LJ[j], (kind, w, j)
            assert pw[w][j][3] or pw[w][j][1] == SW[w] + SJ[j], (kind, w, j)
        keys.append((tuple(pw[w][j][2] for w in range(4)), frozenset(w for w in range(4) if pw[w][j][3])))
    variants = sorted(set(keys), key=lambda k: (k[0], sorted(k[1])))
    return dict(n=n, LJ=LJ, SJ=SJ, LW=LW, SW=SW, var=[variants.index(k) for k in keys], variants=variants)


class Gen:
    KA = dict(dy=0, x=8, partial=16, tps=24, ntiles=28, size=64)

    def __init__(self, c: WCfg):
        self.c = c
        self.out = []
        self.nlabel = 0
        self.S = Alloc("s", 4, 100)
        self.V = Alloc("v", 1, 256)

    def e(self, s, comment=None):
        self.out.append("\t" + s + ("\t; " + comment if comment else ""))

    def label(self, name):
        self.out.append(name + ":")

    def newlabel(self, stem):
        self.nlabel += 1
        return "L_%s_%d" % (stem, self.nlabel)

    def comment(self, s):
        self.out.append("\t; " + s)

    def sel_w(self, dst, vals):
        """dst (SGPR) = vals[wave]"""
        e = self.e
        e("s_mov_b32 %s, %d" % (R("s", dst), vals[0] & 0xFFFFFFFF))
        for w in range(1, 4):
            if vals[w] != vals[0]:
                e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_w), w))
                e("s_cselect_b32 %s, %d, %s" % (R("s", dst), vals[w] & 0xFFFFFFFF, R("s", dst)))

    # -----------------------------------------------------------------------------------------------------------------
    def gen(self):
        c, S, V = self.c, self.S, self.V
        self.plan = {k: piece_plan(c, k) for k in ("d", "x")}
        self.s_split, self.s_pair = 2, 3   # grid = (splits, channel-tile pairs): the workgroups of one split share their pixels' bytes and
        # land on the same XCD (workgroup id % 8), so the sharing happens in that XCD's L2
        self.srd = {"d": S.get(4, 4), "x": S.get(4, 4)}
        self.srdP = S.get(4, 4)
        self.s_ka = S.get(8, 4)
        (self.s_w, self.s_tile, self.s_tend, self.s_cnt, self.s_t0, self.s_t1, self.s_t2, self.s_t3, self.s_ci, self.s_co) = [S.get() for _ in range(10)]
        self.s_lds = {"d": S.get(), "x": S.get()}
        self.s_srcw = {"d": S.get(), "x": S.get()}
        self.s_tsrc = {"d": S.get(), "x": S.get()}
        self.vD_rd = [[V.get() for n in range(4)] for b in range(2)]
        self.vX_rd = [[V.get() for kx in range(3)] for b in range(2)]
        self.v_dma = {k: [V.get() for _ in self.plan[k]["variants"]] for k in ("d", "x")}
        self.v_tmp = [V.get(), V.get()]
        self.v_out = V.get()
        # x window (P = 16 or 64): tap (ky, kx) at k-step s reads positions 32 s + ky P + kx ... — whole half fragments (16 positions) of
        # the SAME position stream per kx, so the stream of a tile is read ONCE into NH consecutive register pairs per kx and the 3 ky taps
        # of a step are three overlapping 4-register windows of it: 6 + 8 transposed reads per k-step instead of 18 + 8.  (The transposed
        # read occupies the LDS pipe like a 64-byte-per-clock access: with 26 per step the kernel is bound by it)
        self.HS = c.P // 16
        self.NH = 2 * c.KS + 2 * self.HS
        self.xwin = c.P in (16, 64) and self.NH == 16 and not getattr(c, "noxwin", False)
        self.F = []
        for s in range(2):
            fd = V.get(16, 4)
            fx = None if self.xwin else V.get(36, 4)
            self.F.append((fd, fx))
        if self.xwin:
            self.XW = [[V.get(2 * self.NH, 4) for kx in range(3)] for b in range(2)]
            self.v_t = [self.XW[1][0] + i for i in range(12)]   # prologue temporaries: buffer 1's window is first written by the main loop
        else:
            self.v_t = [self.F[1][1] + i for i in range(12)]   # prologue temporaries: fragment set 1 is first written by the main loop
        self.nvgpr = V.n
        self.accum_offset = (self.nvgpr + 7) // 8 * 8
        self.nagpr = 144
        self.tmp_i = 0
        self.prologue()
        self.mainloop()
        self.epilogue()
        return self.finish()

    # -----------------------------------------------------------------------------------------------------------------
    def tile_setup(self):
        """descriptors and source row offsets of tile s_tile (both operands); scalar only"""
        c, e = self.c, self.e
        t0, t1, t2 = self.s_t0, self.s_t1, self.s_t2
        out = []
        a = out.append
        tpw = c.TPW
        if tpw > 1 and tpw & (tpw - 1) == 0:
            a("s_lshr_b32 %s, %s, %d" % (R("s", t2), R("s", self.s_tile), tpw.bit_length() - 1))
            a("s_and_b32 %s, %s, %d" % (R("s", self.s_t3), R("s", self.s_tile), tpw - 1))
        elif tpw > 1:
            magic = ((1 << 32) + tpw - 1) // tpw
            a("s_mul_hi_u32 %s, %s, 0x%x" % (R("s", t2), R("s", self.s_tile), magic))  # tile / TPW (exact far beyond any tile count here)
            a("s_mul_i32 %s, %s, %d" % (R("s", self.s_t3), R("s", t2), tpw))
            a("s_sub_u32 %s, %s, %s" % (R("s", self.s_t3), R("s", self.s_tile), R("s", self.s_t3)))
        else:
            a("s_mov_b32 %s, %s" % (R("s", t2), R("s", self.s_tile)))
        for k, ch, ptr, tile_ch in (("d", c.CO, 0, self.s_co), ("x", c.C, 2, self.s_ci)):
            win = c.WIN_PIX * ch * 2
            srd = self.srd[k]
            a("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", t2), win))
            a("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", t2), win))
            a("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", tile_ch)))
            a("s_addc_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
            a("s_add_u32 %s, %s, %s" % (R("s", srd), R("s", self.s_ka + ptr), R("s", t0)))
            a("s_addc_u32 %s, %s, %s" % (R("s", srd + 1), R("s", self.s_ka + ptr + 1), R("s", t1)))
            a("s_and_b32 %s, %s, 0xffff" % (R("s", srd + 1), R("s", srd + 1)))
            if tpw > 1:
                a("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_t3), c.DR * c.W * ch * 2))
                a("s_add_u32 %s, %s, %s" % (R("s", self.s_tsrc[k]), R("s", self.s_srcw[k]), R("s", t0)))
        return out

    def piece_insts(self, k, j, buf):
        c = self.c
        pl = self.plan[k]
        vt = self.v_tmp[self.tmp_i & 1]
        self.tmp_i += 1
        src = self.s_tsrc[k] if c.TPW > 1 else self.s_srcw[k]
        return ["s_add_u32 m0, %s, %d" % (R("s", self.s_lds[k]), pl["LJ"][j] + buf * c.BUF),
                "s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", src), pl["SJ"][j] & 0xFFFFFFFF),
                "v_add_u32 %s, %s, %s" % (R("v", vt), R("s", self.s_t0), R("v", self.v_dma[k][pl["var"][j]])),
                "buffer_load_dwordx4 %s, %s, 0 offen lds" % (R("v", vt), R("s", self.srd[k], 4))]

    def all_pieces(self, buf):
        out = []
        nd, nx = self.plan["d"]["n"], self.plan["x"]["n"]
        for j in range(max(nd, nx)):
            if j < nd:
                out.append(self.piece_insts("d", j, buf))
            if j < nx:
                out.append(self.piece_insts("x", j, buf))
        return out

    def next_tile_insts(self):
        """s_tile = min(s_tile + 1, s_tend)"""
        return ["s_add_u32 %s, %s, 1" % (R("s", self.s_tile), R("s", self.s_tile)),
                "s_min_u32 %s, %s, %s" % (R("s", self.s_tile), R("s", self.s_tile), R("s", self.s_tend))]

    def x_halves(self, buf, js):
        """x window: half fragments js of the tile in buffer buf, for the three kx streams"""
        out = []
        for j in js:
            for kx in range(3):
                out.append("ds_read_b64_tr_b16 %s, %s offset:%d" % (R("v", self.XW[buf][kx] + 2 * j, 2), R("v", self.vX_rd[buf][kx]), j * 2048))
        return out

    def frag_reads(self, fset, step, buf, first=False):
        """reads that must be done before k-step `step` of the tile in buffer buf can start: its dy fragments and (x window) the half
        fragments that step is the first to use — all of the first step's when `first`"""
        c = self.c
        fd, fx = self.F[fset]
        out = []
        for n in range(4):
            for h in range(2):
                out.append("ds_read_b64_tr_b16 %s, %s offset:%d" % (R("v", fd + 4 * n + 2 * h, 2), R("v", self.vD_rd[buf][n]), step * 4096 + h * 2048))
        if self.xwin:
            hi = 2 * step + 2 * self.HS + 1
            return out + self.x_halves(buf, range(0, hi + 1) if first else (hi - 1, hi))
        for t in range(9):
            ky, kx = divmod(t, 3)
            for h in range(2):
                out.append("ds_read_b64_tr_b16 %s, %s offset:%d" % (R("v", fx + 4 * t + 2 * h, 2), R("v", self.vX_rd[buf][kx]), step * 4096 + h * 2048 + ky * c.P * 128))
        return out

    def mfmas(self, fset, step=0, buf=0):
        fd, fx = self.F[fset]
        out = []
        for t in range(9):
            ky, kx = divmod(t, 3)
            xr = self.XW[buf][kx] + 2 * (2 * step + ky * self.HS) if self.xwin else fx + 4 * t
            for n in range(4):
                acc = (t * 4 + n) * 4
                out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (R("a", acc, 4), R("v", xr, 4), R("v", fd + 4 * n, 4), R("a", acc, 4)))
        return out

    def interleave(self, mf, groups, first=1):
        n, k = len(mf), len(groups)
        slots = {}
        if k:
            span = int((n - first) * getattr(self.c, "rspan", 100) / 100)  # (tuning knob: issue everything within the first rspan % of the MFMAs)
            for j, grp in enumerate(groups):
                pos = first + (j * span) // k
                slots.setdefault(pos, []).extend(grp)
        for i, m in enumerate(mf):
            self.e(m)
            for ins in slots.get(i, []):
                self.e(ins)

    # -----------------------------------------------------------------------------------------------------------------
    def prologue(self):
        c, e = self.c, self.e
        ka = self.s_ka
        v = self.v_t
        t0, t1 = self.s_t0, self.s_t1
        self.comment("---- prologue")
        e("s_load_dwordx8 %s, s[0:1], 0x0" % R("s", ka, 8))     # dy, x, partial, tps, ntiles
        lane = v[0]
        e("v_lshrrev_b32 %s, 6, v0" % R("v", v[1]))
        e("v_and_b32 %s, 63, v0" % R("v", lane))
        e("v_readfirstlane_b32 %s, %s" % (R("s", self.s_w), R("v", v[1])))
        # ---- zero the whole LDS allocation (halo rows, right padding, the spill positions): wave w takes 1 KiB blocks w, w+4, ...
        z = self.F[0][0]
        for i in range(4):
            e("v_mov_b32 %s, 0" % R("v", z + i))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", v[2]), R("v", lane)))
        e("s_lshl_b32 %s, %s, 10" % (R("s", t0), R("s", self.s_w)))
        e("v_add_u32 %s, %s, %s" % (R("v", v[2]), R("s", t0), R("v", v[2])))
        nblk = c.LDS // 1024
        for k in range((nblk + 3) // 4):
            off = k * 4096
            if off and off % 61440 == 0:
                e("v_add_u32 %s, %d, %s" % (R("v", v[2]), 61440, R("v", v[2])))
            # the last group may be short: blocks past the allocation are skipped by clamping the address to the wave's last block
            if (k * 4 + 3) >= nblk:
                last = [(k * 4 + w) if (k * 4 + w) < nblk else (k * 4 + w - 4) for w in range(4)]
                self.sel_w(t1, [b * 1024 for b in last])
                e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", v[3]), R("v", lane), R("s", t1)))
                e("ds_write_b128 %s, %s" % (R("v", v[3]), R("v", z, 4)))
            else:
                e("ds_write_b128 %s, %s offset:%d" % (R("v", v[2]), R("v", z, 4), off % 61440))
        # ---- lane parts of the LDS-DMA pieces: position-in-block = lane >> 3, chunk = (lane & 7) ^ sw(lane >> 3)
        l3, l7, ch16, x, off = v[3], v[4], v[5], v[6], v[7]
        e("v_lshrrev_b32 %s, 3, %s" % (R("v", l3), R("v", lane)))
        e("v_and_b32 %s, 7, %s" % (R("v", l7), R("v", lane)))
        e("v_and_b32 %s, 6, %s" % (R("v", ch16), R("v", l3)), "((row >> 1) & 3) * 2 = row & 6")
        e("v_xor_b32 %s, %s, %s" % (R("v", ch16), R("v", l7), R("v", ch16)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", ch16), R("v", ch16)))
        for k, chn in (("d", c.CO), ("x", c.C)):
            pl = self.plan[k]
            for vi, (xbs, deadset) in enumerate(pl["variants"]):
                dst = self.v_dma[k][vi]
                self.sel_w(t0, [xb * 8 - (1 if k == "x" else 0) for xb in xbs])
                e("v_add_u32 %s, %s, %s" % (R("v", x), R("s", t0), R("v", l3)), "pixel column of this lane's position")
                e("v_mov_b32 %s, %d" % (R("v", off), chn * 2))
                e("v_mad_u32_u24 %s, %s, %s, %s" % (R("v", off), R("v", x), R("v", off), R("v", ch16)))
                e("v_cmp_gt_u32 vcc, %d, %s" % (c.W, R("v", x)), "0 <= column < W")
                e("v_mov_b32 %s, 0x80000000" % R("v", x))
                e("v_cndmask_b32 %s, %s, %s, vcc" % (R("v", dst), R("v", x), R("v", off)))
                if deadset:
                    self.sel_w(t1, [1 if w in deadset else 0 for w in range(4)])
                    e("s_cmp_eq_u32 %s, 1" % R("s", t1))
                    e("s_cselect_b64 vcc, -1, 0")
                    e("v_cndmask_b32 %s, %s, %s, vcc" % (R("v", dst), R("v", dst), R("v", x)))
            self.sel_w(self.s_lds[k], pl["LW"])
            self.sel_w(self.s_srcw[k], pl["SW"])
        e("s_waitcnt lgkmcnt(0)")
        # ---- tile pair, split
        nci = c.NCI
        assert nci & (nci - 1) == 0
        e("s_and_b32 %s, %s, %d" % (R("s", self.s_ci), R("s", self.s_pair), nci - 1))
        e("s_lshr_b32 %s, %s, %d" % (R("s", self.s_co), R("s", self.s_pair), nci.bit_length() - 1))
        # partial slab: base + split*n*4 + (co_t*64*9*C + ci_t*64)*4 ; wave part 16w*4 in the lane offset
        nslab = c.CO * 9 * c.C * 4
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_split), nslab))
        e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_split), nslab))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t2), R("s", self.s_co), 64 * 9 * c.C * 4))
        e("s_lshl_b32 %s, %s, 8" % (R("s", self.s_t3), R("s", self.s_ci)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_t2), R("s", self.s_t2), R("s", self.s_t3)))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", self.s_t2)))
        e("s_addc_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdP), R("s", ka + 4), R("s", t0)))
        e("s_addc_u32 %s, %s, %s" % (R("s", self.srdP + 1), R("s", ka + 5), R("s", t1)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdP + 1), R("s", self.srdP + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdP + 2), 64 * 9 * c.C * 4))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdP + 3))
        # channel tile byte offsets (added to the descriptor bases)
        e("s_lshl_b32 %s, %s, 7" % (R("s", self.s_ci), R("s", self.s_ci)))
        e("s_lshl_b32 %s, %s, 7" % (R("s", self.s_co), R("s", self.s_co)))
        for k, chn in (("d", c.CO), ("x", c.C)):
            srd = self.srd[k]
            e("s_sub_u32 %s, %d, %s" % (R("s", srd + 2), c.WIN_PIX * chn * 2, R("s", self.s_co if k == "d" else self.s_ci)), "window bytes behind the channel tile offset")
            e("s_mov_b32 %s, 0x00020000" % R("s", srd + 3))
        # first tile of this split, last tile
        e("s_mul_i32 %s, %s, %s" % (R("s", self.s_tile), R("s", self.s_split), R("s", ka + 6)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_tend), R("s", self.s_tile), R("s", ka + 6)))
        e("s_min_u32 %s, %s, %s" % (R("s", self.s_tend), R("s", self.s_tend), R("s", ka + 7)))
        e("s_sub_u32 %s, %s, %s" % (R("s", self.s_cnt), R("s", self.s_tend), R("s", self.s_tile)), "tiles of this workgroup (>= 1)")
        e("s_sub_u32 %s, %s, 1" % (R("s", self.s_tend), R("s", self.s_tend)))
        # lane offset of the slab stores: ((lane & 15) * 9 * C + 16 w + 4 (lane >> 4)) * 4
        e("v_and_b32 %s, 15, %s" % (R("v", v[8]), R("v", lane)))
        e("v_lshrrev_b32 %s, 4, %s" % (R("v", v[9]), R("v", lane)))
        e("v_mov_b32 %s, %d" % (R("v", v[10]), 9 * c.C * 4))
        e("v_mul_lo_u32 %s, %s, %s" % (R("v", v[8]), R("v", v[8]), R("v", v[10])))
        e("s_lshl_b32 %s, %s, 6" % (R("s", t0), R("s", self.s_w)))
        e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", v[8]), R("v", v[9]), R("v", v[8])))
        e("v_add_u32 %s, %s, %s" % (R("v", self.v_out), R("s", t0), R("v", v[8])))
        # ---- transposed-read bases: lane group g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3; row = 4g + q
        row, p8, cc, rk = v[3], v[4], v[5], v[6]
        e("v_lshrrev_b32 %s, 2, %s" % (R("v", row), R("v", lane)), "4g + q = lane >> 2")
        e("v_and_b32 %s, 3, %s" % (R("v", p8), R("v", lane)))
        e("v_lshlrev_b32 %s, 3, %s" % (R("v", p8), R("v", p8)))
        e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", p8), R("v", row), R("v", p8)), "row*128 + 8p")
        e("v_and_b32 %s, 6, %s" % (R("v", cc), R("v", row)), "sw(row)")
        for n in range(4):
            e("v_xor_b32 %s, %d, %s" % (R("v", rk), 2 * n, R("v", cc)))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", self.vD_rd[0][n]), R("v", rk), R("v", p8)))
            e("v_add_u32 %s, %d, %s" % (R("v", self.vD_rd[1][n]), c.BUF, R("v", self.vD_rd[0][n])))
        e("s_lshl_b32 %s, %s, 1" % (R("s", t0), R("s", self.s_w)), "chunk pair of this wave's 16 input channels")
        for kx in range(3):
            e("v_add_u32 %s, %d, %s" % (R("v", rk), kx, R("v", row)))
            e("v_and_b32 %s, 6, %s" % (R("v", cc), R("v", rk)))
            e("v_xor_b32 %s, %s, %s" % (R("v", cc), R("s", t0), R("v", cc)))
            e("v_lshlrev_b32 %s, 4, %s" % (R("v", cc), R("v", cc)))
            e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", cc), R("v", rk), R("v", cc)))
            e("v_and_b32 %s, 3, %s" % (R("v", rk), R("v", lane)))
            e("v_lshl_add_u32 %s, %s, 3, %s" % (R("v", cc), R("v", rk), R("v", cc)))
            e("v_add_u32 %s, %d, %s" % (R("v", self.vX_rd[0][kx]), c.DYB, R("v", cc)))
            e("v_add_u32 %s, %d, %s" % (R("v", self.vX_rd[1][kx]), c.DYB + c.BUF, R("v", cc)))
        for i in range(self.nagpr):
            e("v_accvgpr_write_b32 a%d, 0" % i)
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier", "LDS is zero everywhere")
        # ---- tile 0 into buffer 0, then tile 1 into buffer 1
        for ins in self.tile_setup():
            e(ins)
        for grp in self.all_pieces(0):
            for ins in grp:
                e(ins)
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        for ins in self.next_tile_insts() + self.tile_setup():
            e(ins)
        ps = self.all_pieces(1)
        for grp in ps[:(len(ps) + 1) // 2]:   # (the second half is issued by the first k-step of the main loop)
            for ins in grp:
                e(ins)
        for ins in self.frag_reads(0, 0, 0, first=True):
            e(ins)

    def mainloop(self):
        c, e = self.c, self.e
        KS = c.KS
        assert KS % 2 == 1 or True
        self.comment("---- main loop: 2 tiles per trip (the LDS buffers alternate), %d k-steps of 36 MFMAs per tile" % KS)
        top, done = self.newlabel("loop"), self.newlabel("done")
        self.label(top)
        fset = 0
        npieces = self.plan["d"]["n"] + self.plan["x"]["n"]
        for b in range(2):
            carry = getattr(self, "_carry", None)
            for s in range(KS):
                self.comment("buffer %d k-step %d" % (b, s))
                last = s == KS - 1
                e("s_waitcnt lgkmcnt(0)")
                if last:
                    # the tile in the other buffer has landed for every wave; this buffer's fragments have all been read
                    e("s_waitcnt vmcnt(0)")
                    e("s_barrier")
                    groups = [[r] for r in self.frag_reads(fset ^ 1, 0, b ^ 1, first=True)]
                    ps = self.all_pieces(b)
                    half = (len(ps) + 1) // 2
                    setup = self.next_tile_insts() + self.tile_setup()
                    first = [setup + ps[0]] + ps[1:half]
                    self._carry_next = ps[half:]
                    groups = self.merge(groups, first)
                else:
                    groups = [[r] for r in self.frag_reads(fset ^ 1, s + 1, b)]
                    if s == 0:
                        # second half of the pieces issued behind the previous tile's barrier (static: both buffers emit the same count)
                        ps = self.all_pieces(b ^ 1)
                        half = (len(ps) + 1) // 2
                        groups = self.merge(groups, ps[half:])
                self.interleave(self.mfmas(fset, s, b), groups)
                fset ^= 1
            e("s_sub_u32 %s, %s, 1" % (R("s", self.s_cnt), R("s", self.s_cnt)))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_cnt))
            if b == 0:
                e("s_cbranch_scc1 %s" % done)
            else:
                e("s_cbranch_scc0 %s" % top)
        assert (2 * KS) % 2 == 0  # the fragment set parity is the same at the top of every trip
        self.label(done)

    @staticmethod
    def merge(a, b):
        if not b:
            return a
        out = []
        na, nb = len(a), len(b)
        ib = 0
        for i, g in enumerate(a):
            out.append(g)
            while ib < nb and (ib + 1) * na <= (i + 1) * nb:
                out.append(b[ib])
                ib += 1
        out.extend(b[ib:])
        return out

    def epilogue(self):
        c, e = self.c, self.e
        self.comment("---- epilogue: 36 accumulator tiles -> this split's slab")
        e("s_waitcnt vmcnt(0)", "the look-ahead pieces (never used) have landed: no LDS-DMA is in flight when the wave ends")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 15")
        for t in range(9):
            for n in range(4):
                acc = (t * 4 + n) * 4
                e("s_mov_b32 %s, %d" % (R("s", self.s_t0), (n * 16 * 9 + t) * c.C * 4))
                e("buffer_store_dwordx4 %s, %s, %s, %s offen" % (R("a", acc, 4), R("v", self.v_out), R("s", self.srdP, 4), R("s", self.s_t0)))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")

    def finish(self):
        c = self.c
        name = c.name
        lds = c.LDS
        assert lds <= 160 * 1024
        total_v = self.accum_offset + self.nagpr
        hdr = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6", "\t.text", "\t.protected\t%s" % name,
               "\t.globl\t%s" % name, "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
        tail = ["\t.section\t.rodata,\"a\",@progbits", "\t.p2align\t6, 0x0", "\t.amdhsa_kernel %s" % name]
        kd = dict(group_segment_fixed_size=lds, private_segment_fixed_size=0, kernarg_size=self.KA["size"],
                  user_sgpr_count=2, user_sgpr_dispatch_ptr=0, user_sgpr_queue_ptr=0, user_sgpr_kernarg_segment_ptr=1,
                  user_sgpr_dispatch_id=0, user_sgpr_kernarg_preload_length=0, user_sgpr_kernarg_preload_offset=0,
                  user_sgpr_private_segment_size=0, uses_dynamic_stack=0, enable_private_segment=0,
                  system_sgpr_workgroup_id_x=1, system_sgpr_workgroup_id_y=1, system_sgpr_workgroup_id_z=0,
                  system_sgpr_workgroup_info=0, system_vgpr_workitem_id=0, next_free_vgpr=total_v,
                  next_free_sgpr=self.S.n, accum_offset=self.accum_offset, reserve_vcc=1, float_round_mode_32=0,
                  float_round_mode_16_64=0, float_denorm_mode_32=3, float_denorm_mode_16_64=3, dx10_clamp=1, ieee_mode=1,
                  fp16_overflow=0, tg_split=0)
        for k, v in kd.items():
            tail.append("\t\t.amdhsa_%s %d" % (k, v))
        tail += ["\t.end_amdhsa_kernel", "\t.text", "\t.amdgpu_metadata", "---", "amdhsa.kernels:", "  - .agpr_count:     %d" % self.nagpr, "    .args:"]
        off = 0
        for i in range(3):
            tail.append("      - .address_space:  global\n        .offset:         %d\n        .size:           8\n        .value_kind:     global_buffer" % off)
            off += 8
        for i in range((self.KA["size"] - off) // 4):
            tail.append("      - .offset:         %d\n        .size:           4\n        .value_kind:     by_value" % off)
            off += 4
        tail += ["    .group_segment_fixed_size: %d" % lds, "    .kernarg_segment_align: 8", "    .kernarg_segment_size: %d" % self.KA["size"],
                 "    .max_flat_workgroup_size: 256", "    .name:           %s" % name, "    .private_segment_fixed_size: 0",
                 "    .sgpr_count:     %d" % (self.S.n + 6), "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % name,
                 "    .uniform_work_group_size: 1", "    .uses_dynamic_stack: false", "    .vgpr_count:     %d" % total_v,
                 "    .vgpr_spill_count: 0", "    .wavefront_size: 64", "amdhsa.target:   amdgcn-amd-amdhsa--gfx950",
                 "amdhsa.version:\n  - 1\n  - 2", "...", "\t.end_amdgpu_metadata"]
        body = self.out + ["\t.p2align 8", ".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name)]
        self.lds_bytes = lds
        return "\n".join(hdr + body + tail) + "\n"


# ---------------------------------------------------------------------------------------------------------------------
VARIANTS = {
    # ResNet-50 at 224 px: conv2 of the bottlenecks of layer 3 (14 x 14, 256 -> 256), layer 2 (28 x 28, 128 -> 128), layer 4 (7 x 7, 512 -> 512),
    # layer 1 (56 x 56, 64 -> 64: two-row tiles, 28 per image)
    "wg3_l3": WCfg("wg3_l3", H=14, W=14, P=16, C=256, CO=256, geom="img", DR=14),
    "wg3_l2": WCfg("wg3_l2", H=28, W=28, P=32, C=128, CO=128, geom="rows", DR=7),
    "wg3_l4": WCfg("wg3_l4", H=7, W=7, P=8, C=512, CO=512, geom="pack", DR=32, IPT=4),
    "wg3_l1": WCfg("wg3_l1", H=56, W=56, P=64, C=64, CO=64, geom="rows", DR=2),
    # BResNet-50's deep stem (configs[3]): the 3x3 convolutions at 112 x 112 with channels padded to 64 (one-row tiles, 112 per image)
    "wg3_s112": WCfg("wg3_s112", H=112, W=112, P=128, C=64, CO=64, geom="rows", DR=1),
    # the same layers at the other sizes of the progressive-resize recipe (BASELINE configs[4]): a = 160 px (40 / 20 / 10), b = 320 px
    # (80 / 40 / 20 / 10).  (Layer 4 at 160 px, 5 x 5, stays on the implicit-GEMM kernel: its packed tile would be 61 % padding.)
    "wg3_l1a": WCfg("wg3_l1a", H=40, W=40, P=48, C=64, CO=64, geom="rows", DR=4),
    "wg3_l2a": WCfg("wg3_l2a", H=20, W=20, P=32, C=128, CO=128, geom="rows", DR=5),
    "wg3_l3a": WCfg("wg3_l3a", H=10, W=10, P=16, C=256, CO=256, geom="img", DR=10),
    "wg3_l1b": WCfg("wg3_l1b", H=80, W=80, P=96, C=64, CO=64, geom="rows", DR=2),
    "wg3_l2b": WCfg("wg3_l2b", H=40, W=40, P=48, C=128, CO=128, geom="rows", DR=4),
    "wg3_l3b": WCfg("wg3_l3b", H=20, W=20, P=32, C=256, CO=256, geom="rows", DR=5),
    "wg3_l4b": WCfg("wg3_l4b", H=10, W=10, P=16, C=512, CO=512, geom="img", DR=10),
    # BResNet-50 (configs[3]): conv2 of a striding block runs at the block's INPUT resolution (the stride sits in the blur pool behind it)
    "wg3_v2": WCfg("wg3_v2", H=56, W=56, P=64, C=128, CO=128, geom="rows", DR=2),
    "wg3_v3": WCfg("wg3_v3", H=28, W=28, P=32, C=256, CO=256, geom="rows", DR=7),
    "wg3_v4": WCfg("wg3_v4", H=14, W=14, P=16, C=512, CO=512, geom="img", DR=14),
}


def generate(name, **over):
    c = VARIANTS[name]
    if over:
        c = WCfg(**{**c.__dict__, **over})
    g = Gen(c)
    return c, g, g.gen()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for name in VARIANTS:
        c, g, text = generate(name)
        with open(os.path.join(a.out, name + ".s"), "w") as f:
            f.write(text)
        print("%s: %d vgpr + %d agpr, lds %d, %d lines" % (name, g.accum_offset, g.nagpr, g.lds_bytes, text.count("\n")))


if __name__ == "__main__":
    main()
