#!/usr/bin/env python3
"""dconv_gen.py — generator of the hand-scheduled gfx950 (MI355X) direct 3x3 / stride-1 convolution kernels.

What the kernels replace: the cuDNN conv forward / data-gradient under `model(data)` / `loss.backward()` of the reference
(call form /root/reference/sota_imagenet/callbacks.py:316-317) for the 3x3 convolutions of ResNet-50's layers 3 and 4 —
the same IgemmArgs contract as conv_igemm8.hip (BN statistics / BN-backward sums in the epilogue), selected in launch_igemm().

Why a generator that prints assembly: the structure below needs ONE wave per SIMD with 224 accumulator registers, fragment
reads of the next k-step and LDS-DMA issue placed BETWEEN the MFMAs of the current one, and counted waits across one
barrier per 112 MFMAs.  hipcc does not keep such a schedule; the text printed here is the schedule.

Structure (one workgroup = 4 waves = one wave per SIMD, one tile per workgroup):
  tile        RO output rows x P positions per row (P = padded row pitch, a multiple of 8, >= W + 1) of IPT whole images
              x BN = 256 output channels; waves 2 (M) x 2 (N); a wave owns MFR x 8 accumulator tiles of 16 x 16
              (v_mfma_f32_16x16x32_bf16, operands swapped: D^T = W * A^T, a lane ends with 4 channels of one pixel).
  A operand   the zero-haloed input tile of ONE 64-channel chunk is staged ONCE ([position][128 B] rows, LDS-DMA) and
              all 9 taps read it at shifted addresses: output position p, tap (ky, kx) reads position p + ky*P + kx, so
              a fragment (16 consecutive positions) of any tap is again 16 consecutive rows.  The 16-byte chunk index is
              rotated by (position & ~1) on the DMA source and on the read: conflict-free for ds_read_b128 at EVERY shift
              (the XOR swizzle of the other kernels is conflict-free only at even shifts), and because P % 8 == 0 the
              ky shift and the fragment index are immediates of the read: 6 address registers serve every A read.
              Staging bytes per 64-channel chunk: 1 A tile + 9 weight slabs instead of 9 + 9.
  B operand   weight rows [256][128 B] per (chunk, tap) stage, 2-stage ring, rows permuted so that a lane's accumulators
              of a tile pair are 8 consecutive channels (16-byte stores, no lane exchange).
  schedule    per stage 2 substeps of 56 MFMAs; substep s computes on fragment set s & 1 while the 15 ds_read_b128 of
              substep s + 1 and the LDS-DMA pieces of stage + 2 are issued between its MFMAs; ONE s_barrier per stage
              (at the start of its second substep) behind counted vmcnt / lgkmcnt(0).
  epilogue    accumulators -> bf16 -> 16-byte stores; BN statistics (sum, sum of squares of the rounded values) or the
              BN-backward sums of the layer whose activation gradient the output is, reduced by DPP row sums, one partial
              row per workgroup.
"""
import argparse
import os
import sys
from dataclasses import dataclass


# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class Cfg:
    name: str
    H: int
    W: int
    P: int            # LDS row pitch in positions
    IPT: int          # images per tile
    Cin: int          # channels of the input tensor (= pixel stride = reduction per tap)
    NCOLS: int        # columns of the output tensor (multiple of 256; blockIdx.y picks the 256-column tile)
    stats: int        # 0 none, 1 BN statistics of the output, 2 BN-backward sums (ReLU mask), 3 BN-backward sums under a LEAKY ReLU mask: dz = dx where the bit is set,
                      # dx * 0.01 elsewhere (the slope of BASELINE configs[3]'s activation, /root/reference/configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51)
    skew: int = 0     # s_nop-based wave stagger after each barrier (experiment knob)
    probe: int = 0    # timing probes (WRONG results): 1 no LDS-DMA in the main loop, 2 no fragment reads, 4 no barriers,
                      # 8 no BN-backward input loads, 16 no statistics arithmetic, 32 no output stores
    WM: int = 2       # waves along the pixel dimension
    WN: int = 2       # waves along the output channels (WM * WN = 4: one wave per SIMD)
    NT: int = 8       # 16-column tiles per wave
    NB: int = 2       # stages of the weight ring (3 where LDS allows: the pieces of stage s + 3 then have two stages to land)
    ROWS_T: int = 0   # 0: a tile is IPT whole images; else a tile is ROWS_T consecutive output rows of ONE image (H % ROWS_T == 0)
    bnin: int = 0     # 1: the input is the RAW output y of the previous convolution: its BatchNorm + ReLU (a = relu(y * scale[c] + shift[c]), what
                      # bn_apply_kernel computes) is applied to the staged tile in LDS before the taps read it, and the kernel leaves a and its ReLU
                      # bit mask in memory as a by-product (every later reader of a — the weight gradient, the BN backward — is unchanged); forward only
    fp8: int = 0      # 1: e4m3 operands (input tensor and weights: one byte per element), bf16 output = accumulator * oscale / (scale_in * scale_wt): a chunk
                      # is 128 channels in the SAME 128-byte LDS rows, the two 16-byte reads per lane that feed two 16x16x32 bf16 MFMAs are the 32-byte
                      # operand of ONE v_mfma_f32_16x16x128_f8f6f4 (k is a summation index: the same bytes on both sides) — the stage of a tap does twice
                      # the channels in the same MFMA time (tools/micro/mfma_shape_random.hip: 4.06 against 2.02 PFLOP/s on random data)
    s2d: int = 0      # 1: the data gradient of a 3x3 / stride-2 convolution: H x W is the dy image the tile stages, the output is 2H x 2W; four
                      # kernel classes (output parities, Gen.class_stages), workgroup id y = class * column tiles + column tile

    @property
    def ES(self):     # bytes per input / weight element
        return 1 if self.fp8 else 2

    @property
    def CH(self):     # channels per chunk (= one 128-byte LDS row)
        return 128 // self.ES

    @property
    def BN(self):     # output columns per workgroup
        return self.WN * self.NT * 16

    @property
    def TPI(self):    # tiles per image
        return self.H // self.ROWS_T if self.ROWS_T else 1

    @property
    def ABASE(self):
        return 0

    @property
    def NA(self):     # A buffers: the tile of chunk c + 1 lands while chunk c is computed; one chunk (Cin = 64) needs one buffer
        return 1 if self.Cin == self.CH else 2

    @property
    def BBASE(self):
        return self.NA * self.ABUF

    @property
    def SR(self):     # LDS rows per image slot (the halo row between two images is shared)
        return self.H + 1 if self.IPT > 1 else self.H + 2

    @property
    def RO(self):     # output rows enumerated per tile (IPT > 1: includes one garbage row per image)
        if self.ROWS_T:
            return self.ROWS_T
        return self.IPT * self.SR if self.IPT > 1 else self.H

    @property
    def NFRAG(self):
        assert (self.RO * self.P) % 16 == 0
        return self.RO * self.P // 16

    @property
    def MFR(self):
        assert self.NFRAG % self.WM == 0
        return self.NFRAG // self.WM

    @property
    def LROWS(self):  # LDS rows of the A tile
        return self.RO + 2

    @property
    def ABUF(self):   # bytes of one A buffer (the 2 positions the last garbage columns read past it fall into what follows it in LDS)
        return self.LROWS * self.P * 128

    @property
    def ASTRIDE(self):
        return self.ABUF

    @property
    def BSTAGE(self):
        return self.BN * 128

    @property
    def NPB(self):    # weight pieces per wave and stage
        return self.BSTAGE // 1024 // 4

    @property
    def w_row(self):  # bytes of one weight row [taps][Cin]
        return 9 * self.Cin * self.ES

    @property
    def tile_rows(self):
        return self.ROWS_T if self.ROWS_T else self.IPT * self.H


# LDS: [A buffer 0][A buffer 1][weight ring NB stages]; the statistics scratch reuses the ring


LEAKY_BITS = 0x3c23d70a   # 0.01f: the slope of the stats == 3 epilogues
NCLS = 3  # tile classes of a row tile: 0 first of its image (the row above is zero halo), 1 middle, 2 last (the row below is)


def a_rows(c, cls):
    """LDS rows of the A tile of a tile of class `cls`: list of (g, source row constant in bytes | None = stays zero).  Row tiles
    address a window that starts one image row above the tile (srdA), so the constant does not depend on the tile."""
    rows = []
    for g in range(c.LROWS):
        if c.ROWS_T:
            dead = (cls == 0 and g == 0) or (cls == 2 and g == c.LROWS - 1) or (c.TPI == 1 and g in (0, c.LROWS - 1))
            rows.append((g, None if dead else g * c.W * c.Cin * c.ES))
        elif c.IPT == 1:
            rows.append((g, (g - 1) * c.W * c.Cin * c.ES if 1 <= g <= c.H else None))
        else:
            i, y = divmod(g - 1, c.SR) if g >= 1 else (0, -1)
            ok = g >= 1 and y < c.H and i < c.IPT
            rows.append((g, ((i * c.H + y) * c.W) * c.Cin * c.ES if ok else None))
    return rows


def tile_classes(c):
    if not c.ROWS_T or c.TPI == 1:
        return [0]
    return [0, 2] if c.TPI == 2 else [0, 1, 2]


def live_rows(c):
    """rows that hold image data in SOME tile parity"""
    return sorted({g for cls in tile_classes(c) for g, src in a_rows(c, cls) if src is not None})


def a_slots(c):
    """the A pieces of a chunk as slots: (variant = 8-position block index in the row, [LDS row for wave 0..3 | None])"""
    live = live_rows(c)
    slots = []
    for xb in range(c.P // 8):
        if xb * 8 > c.W:   # blocks entirely in the right padding of the row stay zero
            continue
        for k in range(0, len(live), 4):
            grp = live[k:k + 4]
            slots.append((xb, grp + [None] * (4 - len(grp))))
    return slots


def written_blocks(c):
    bpr = c.P // 8
    return {(g * bpr + xb) * 1024 for xb, grp in a_slots(c) for g in grp if g is not None}


def b_piece_const(c, w, i):
    """source constant (bytes) of B piece i of wave w: the first channel of its 8 rows"""
    R0 = (c.BN // 4) * w + 8 * i
    wn, rem = divmod(R0, c.NT * 16)
    n = rem // 16
    p, odd = n >> 1, n & 1
    cout_base = wn * (c.NT * 16) + p * 32 + 4 * odd
    return cout_base * c.w_row


def tables(c):
    """[tile class 0 / 1 / 2][wave] -> 64 words: [A LDS offsets][A source constants][B source constants].  A slot whose row holds no
    data for this wave / class repeats one of the wave's own pieces of the same variant (same bytes to the same place)."""
    sl = a_slots(c)
    bpr = c.P // 8
    out = []
    for cls in range(NCLS):
        src = dict(a_rows(c, cls if cls in tile_classes(c) else tile_classes(c)[0]))
        rows = []
        for w in range(4):
            lds, srcs = [], []
            for xb, grp in sl:
                g = grp[w]
                if g is None or src[g] is None:
                    cands = [gg[w] for x2, gg in sl if x2 == xb and gg[w] is not None and src[gg[w]] is not None]
                    if not cands:   # (two-row tiles: one row per wave, and this wave's row is halo in this tile class) another wave's piece: the same bytes to the same place
                        cands = [g2 for x2, gg in sl if x2 == xb for g2 in gg if g2 is not None and src[g2] is not None]
                    assert cands, "no piece of variant %d in this tile class" % xb
                    g = cands[0]
                lds.append((g * bpr + xb) * 1024)
                srcs.append(src[g])
            words = lds + srcs + [b_piece_const(c, w, i) for i in range(c.NPB)]
            assert len(words) <= 64
            rows.append(words + [0] * (64 - len(words)))
        out.append(rows)
    return out


TR_SKIP = 0x80000000   # transform-slot flag (ttables): write the block back unchanged, store nothing


def ttables(c):
    """Cfg.bnin: [tile class][wave] -> 64 words: [NPA LDS offsets][NPA source constants | flags] of the blocks this wave TRANSFORMS (BatchNorm + ReLU in
    LDS), slot for slot the pieces of a_slots().  Unlike the LDS-DMA table a slot without a row for this wave, or whose row is zero halo in this tile
    class, is not replaced by a duplicate (a block must be transformed once): it carries TR_SKIP and names one of the wave's own blocks, which is
    read and not written."""
    sl = a_slots(c)
    bpr = c.P // 8
    out = []
    for cls in range(NCLS):
        src = dict(a_rows(c, cls if cls in tile_classes(c) else tile_classes(c)[0]))
        rows = []
        for w in range(4):
            lds, srcs = [], []
            for xb, grp in sl:
                g = grp[w]
                if g is None:      # no block of this variant for the wave: any of its own blocks, unchanged
                    g2 = [gg[w] for x2, gg in sl if x2 == xb and gg[w] is not None][0]
                    lds.append((g2 * bpr + xb) * 1024)
                    srcs.append(TR_SKIP)
                elif src[g] is None:   # halo row of this tile class: zeros stay zeros
                    lds.append((g * bpr + xb) * 1024)
                    srcs.append(TR_SKIP)
                else:
                    assert 0 <= src[g] < TR_SKIP // 2
                    lds.append((g * bpr + xb) * 1024)
                    srcs.append(src[g])
            words = lds + srcs
            assert len(words) <= 64
            rows.append(words + [0] * (64 - len(words)))
        out.append(rows)
    return out


# ---------------------------------------------------------------------------------------------------------------------
class Alloc:
    def __init__(self, prefix, first, limit):
        self.p, self.n, self.limit = prefix, first, limit

    def get(self, n=1, align=1):
        self.n = (self.n + align - 1) // align * align
        r = self.n
        self.n += n
        assert self.n <= self.limit, "out of %s registers" % self.p
        return r


def R(p, i, n=1):
    return "%s%d" % (p, i) if n == 1 else "%s[%d:%d]" % (p, i, i + n - 1)


class Gen:
    def __init__(self, c: Cfg):
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

    # -----------------------------------------------------------------------------------------------------------------
    def gen(self):
        c = self.c
        S, V, e = self.S, self.V, self.e
        sl = a_slots(c)
        NPA = len(sl)
        self.NPA = NPA
        self.APS = (NPA + 7) // 8                    # A pieces issued per first substep of a tap
        self.ATAPS = (NPA + self.APS - 1) // self.APS  # taps whose first substep carries A pieces (<= 8)
        assert (self.ATAPS <= 8 or c.NA == 1) and c.WM * c.WN == 4

        # ---- registers ----------------------------------------------------------------------------------------------
        # s[0:1] kernarg, s2 = workgroup id x (tile), s3 = workgroup id y (column tile)
        self.s_tile, self.s_nt = 2, 3
        self.srdA = S.get(4, 4)
        self.srdB = S.get(4, 4)
        self.srdO = S.get(4, 4)
        self.srdX = S.get(4, 4)   # statistics rows
        # the piece table of this wave: in SGPRs (scalar loads, as many as 24 words) or, when larger, in a VGPR read with v_readlane
        self.tab_sgpr = (2 * NPA + c.NPB <= 24) and c.TPI == 1 and not (c.s2d and c.stats >= 2) and not c.bnin and not c.fp8   # (s2d + BN-backward sums, bnin, fp8: out of SGPRs)
        if not self.tab_sgpr:
            self.srdK = S.get(4, 4)   # the piece tables in the kernarg segment
        if c.stats >= 2:
            self.srdM = S.get(4, 4)   # bn_bits
            self.srdY = S.get(4, 4)   # bn_y
            self.srdMu = S.get(4, 4)  # bn_mean
            self.srdIs = S.get(4, 4)  # bn_invstd
        self.s_wt = S.get(9, 4)
        (self.s_cnt, self.s_cC, self.s_cN, self.s_t0, self.s_t1, self.s_w, self.s_wm, self.s_wn, self.s_nch, self.s_ldsBw,
         self.s_stg) = [S.get() for _ in range(11)]
        if self.tab_sgpr:
            self.s_tbl = S.get(24, 4)
        else:
            self.s_a, self.s_b, self.s_par, self.s_img = S.get(), S.get(), S.get(), S.get()
        self.s_ka = S.get(16, 4)         # the 8 pointers
        self.s_kb = S.get(4, 4)
        if c.s2d:
            self.s_cls, self.s_clsoff, self.s_cNN = S.get(), S.get(), S.get()
        if c.fp8:
            self.s_sci = S.get(2, 2)      # scale_in pointer (kernarg slot `rsvd`)
            self.s_x4 = S.get(4, 4)       # kernarg 0x70: -, oscale, scale_wt pointer
        if c.bnin:
            assert c.stats in (0, 1) and not c.s2d
            self.srdA1 = S.get(4, 4)     # a (the normalised input) out: the window of srdA on the other tensor
            self.srdBt = S.get(4, 4)     # its ReLU bits out: 1 byte per 16-byte vector = the same offsets >> 4
            self.srdSS = S.get(4, 4)     # [2][Cin] floats: scale, shift
            self.s_sel = S.get(2, 2)
            self.s_ta, self.s_tb, self.s_tb2, self.s_dead, self.s_last, self.s_k1 = [S.get() for _ in range(6)]

        self.v_tid = 0
        self.vA_rd = [[V.get() for kk in range(2)] for kx in range(3)]
        self.vB_rd = [[V.get() for kk in range(2)] for st in range(c.NB)]
        nvar = c.P // 8
        self.vA_dma = [V.get() for _ in range(nvar)]
        self.vB_dma = [V.get() for _ in range(2)]
        self.v_out = V.get()
        self.v_tab = V.get()             # this wave's piece table: word i in lane i (read with v_readlane)
        self.v_kg = V.get()              # lane >> 4 (kept for the epilogue)
        self.F = []
        if c.fp8:
            # 8-register operands (both 16-byte halves of a lane's 32 bytes): two whole sets of pixel fragments (the next stage's are read under this
            # stage's MFMAs) and four rolling slots for the weight fragments (column n + 2's slot is read while column n computes)
            assert c.NB == 3 and not c.bnin
            self.A8 = [V.get(8 * c.MFR, 4), V.get(8 * c.MFR, 4)]
            self.B8 = [V.get(8, 4) for _ in range(4)]
            # (the epilogue and the prologue use the fragment registers as temporaries)
            self.F = [(self.A8[0], self.A8[0] + 4 * c.MFR), (self.A8[1], self.B8[3] + 8 - 4 * c.NT)]
        for s in range(0 if c.fp8 else 2):
            fa = V.get(4 * c.MFR, 4)
            fb = V.get(4 * c.NT, 4)
            self.F.append((fa, fb))
        # prologue temporaries live in fragment set 1, which is first written by the main loop (v_t[2] = the kept lane >> 4)
        self.v_t = [self.F[1][0] + i for i in range(10)]
        self.v_t[2] = self.v_kg
        if c.stats >= 2:
            # BN-backward sums: y / mask of the same (pixel, 8 channels) vectors as the output in two register sets, mean / invstd of
            # this lane's 8 channels per tile pair.  The sets of pairs 0 and 1 are loaded in the PROLOGUE (the main loop does not
            # touch these registers), pairs 2 and 3 under the arithmetic of pairs 0 and 1.
            if c.fp8:
                # (e4m3: the 8-register operands leave no room for two sets: the second one is loaded at the start of the epilogue INTO the fragment
                # registers, free by then; the epilogue's temporaries use the first 51 of them)
                self.late_pair1 = True
                f0 = self.A8[0]
                free = list(range(f0 + 52, f0 + 16 * c.MFR)) + [r + i for r in self.B8 for i in range(8)]
                free4 = [r for r in free if r % 4 == 0 and all(r + i in free for i in range(4))]
                m1 = next(r for r in free4 if all(r + i in free for i in range(16)))
                y1 = [r for r in free4 if not (m1 <= r < m1 + 16)][:c.MFR]
                assert len(y1) == c.MFR, "no room for the late register set"
                self.ysets = [[V.get(4, 4) for m in range(c.MFR)], y1]
                self.msets = [V.get(16, 4), m1]
            else:
                self.ysets = [[V.get(4, 4) for m in range(c.MFR)] for _ in range(2)]
                self.msets = [V.get(16, 4) for _ in range(2)]   # mean[8], invstd[8]
            self.v_chan = V.get()
            self.alloc_tile_masks()
        if c.fp8:
            self.v_osc = V.get()             # oscale / (scale_in * scale_wt): applied to the accumulators before the one bf16 rounding
        if c.bnin:
            self.v_tab2 = V.get()            # this wave's transform table (ttables): word i in lane i
            self.v_lane16 = V.get()          # lane * 16: a lane transforms the 16 bytes its LDS-DMA lane wrote
            self.v_ssoff = V.get()           # byte offset of this lane's 8 channels in a chunk's 64 floats of scale / shift
            self.v_sc = V.get(8, 4)
            self.v_sh = V.get(8, 4)
            # two register sets, alternating by piece: LDS address, raw data, work / packed result, bits, store offsets
            # (2 x pieces per substep sets: a block is read into one set in the substep before the one that works on it)
            nsets = 2 * (self.tr_plan(9)[2] if c.NA == 2 else 1)
            self.tr = [dict(ta=V.get(), d=V.get(4, 4), f=V.get(8, 4), bits=V.get(), o=V.get(), o2=V.get()) for _ in range(nsets)]
        self.nvgpr = V.n
        self.accum_offset = (self.nvgpr + 7) // 8 * 8
        self.nagpr = c.MFR * c.NT * 4
        assert self.nagpr <= 256

        self.prologue()
        self.mainloop()
        self.epilogue()
        return self.finish()

    # -----------------------------------------------------------------------------------------------------------------
    # kernel arguments: 9 pointers (the 9th unused: reserved), 9 weight-tap byte offsets, the chunk count, padding to 128 bytes,
    # then the per-wave piece tables (3 tile classes x 4 waves x 64 words)
    KA = dict(in_=0, wt=8, out=16, stat=24, bn_y=32, bn_bits=40, bn_mean=48, bn_invstd=56, rsvd=64, wtap=72, nchunks=108,
              table=128, size=128 + NCLS * 4 * 256)

    def never_written_blocks(self):
        """1 KiB blocks (relative to an A buffer's base) that no LDS-DMA piece ever writes: halo rows, right padding, the tail"""
        c = self.c
        written = written_blocks(c)
        return [b * 1024 for b in range(c.ABUF // 1024) if b * 1024 not in written]

    def dynamic_halo_blocks(self):
        """blocks of rows that hold data in one tile parity and must be zero in the other (row tiles of an image)"""
        c = self.c
        bpr = c.P // 8
        dyn = [g for g in live_rows(c) if any(dict(a_rows(c, cls))[g] is None for cls in tile_classes(c))]
        return [(g * bpr + xb) * 1024 for g in dyn for xb in range(bpr) if (g * bpr + xb) * 1024 in written_blocks(c)]

    def prologue(self):
        c, e = self.c, self.e
        ka, kb = self.s_ka, self.s_kb
        v = self.v_t
        t0, t1 = self.s_t0, self.s_t1
        self.comment("---- prologue: kernel arguments, lane constants, descriptors, first loads")
        e("s_load_dwordx16 %s, s[0:1], 0x0" % R("s", ka, 16))       # in, wt, out, stat, bn_y, bn_bits, bn_mean, bn_invstd
        e("s_load_dwordx8 %s, s[0:1], 0x48" % R("s", self.s_wt, 8))  # wtap[0..7]
        e("s_load_dwordx2 %s, s[0:1], 0x68" % R("s", kb + 2, 2))    # wtap[8], nchunks
        if c.fp8:
            e("s_load_dwordx2 %s, s[0:1], 0x40" % R("s", self.s_sci, 2), "scale_in (device scalar) or null")
            e("s_load_dwordx4 %s, s[0:1], 0x70" % R("s", self.s_x4, 4), "-, oscale, scale_wt")
        lane, r, kg = v[0], v[1], v[2]
        e("v_lshrrev_b32 %s, 6, v0" % R("v", v[3]))
        e("v_and_b32 %s, 63, v0" % R("v", lane))
        e("v_readfirstlane_b32 %s, %s" % (R("s", self.s_w), R("v", v[3])))
        e("v_and_b32 %s, 15, v0" % R("v", r))
        e("v_bfe_u32 %s, v0, 4, 2" % R("v", kg))
        e("s_nop 3")
        if c.s2d:
            # workgroup id y = class * column tiles + column tile; the class's first output pixel (ph, pw) as a byte offset into the tile's window
            nct = c.NCOLS // c.BN
            assert nct & (nct - 1) == 0
            e("s_lshr_b32 %s, %s, %d" % (R("s", self.s_cls), R("s", self.s_nt), nct.bit_length() - 1))
            e("s_and_b32 %s, %s, %d" % (R("s", self.s_nt), R("s", self.s_nt), nct - 1))
            offs = [(ph * 2 * c.W + pw) * c.NCOLS * 2 for (ph, pw), _ in self.S2D_CLASSES]
            e("s_mov_b32 %s, %d" % (R("s", self.s_clsoff), offs[0]))
            for k in range(1, 4):
                e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_cls), k))
                e("s_cselect_b32 %s, %d, %s" % (R("s", self.s_clsoff), offs[k], R("s", self.s_clsoff)))
        if self.tab_sgpr:
            # this wave's piece table: kernarg + 128 + w*256, 24 words by scalar loads (they do not wait for the loads above)
            e("s_lshl_b32 %s, %s, 8" % (R("s", t0), R("s", self.s_w)))
            e("s_add_u32 %s, s0, %s" % (R("s", kb), R("s", t0)))
            e("s_addc_u32 %s, s1, 0" % R("s", kb + 1))
            e("s_load_dwordx16 %s, %s, 0x80" % (R("s", self.s_tbl, 16), R("s", kb, 2)))
            e("s_load_dwordx8 %s, %s, 0xc0" % (R("s", self.s_tbl + 16, 8), R("s", kb, 2)))
        else:
            # this wave's piece table: 64 words at kernarg + 128 + (parity*4 + w)*256, word i into lane i (one vector load; the
            # kernarg pointer is in s[0:1] from the start, so this does not wait for the scalar loads above)
            e("s_add_u32 %s, s0, 128" % R("s", self.srdK))
            e("s_addc_u32 %s, s1, 0" % R("s", self.srdK + 1))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdK + 1), R("s", self.srdK + 1)))
            e("s_mov_b32 %s, %d" % (R("s", self.srdK + 2), NCLS * 1024 * (2 if c.bnin else 1)))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdK + 3))
            # tile -> image (s_img), row tile t inside it (s_par), class (first 0 / middle 1 / last 2)
            if c.TPI == 1:
                e("s_mov_b32 %s, %s" % (R("s", self.s_img), R("s", self.s_tile)))
                e("s_mov_b32 %s, 0" % R("s", self.s_par))
            else:
                magic = ((1 << 32) + c.TPI - 1) // c.TPI
                e("s_mul_hi_u32 %s, %s, 0x%x" % (R("s", self.s_img), R("s", self.s_tile), magic), "tile / TPI (exact for tiles < 2^32 / TPI)")
                e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_img), c.TPI))
                e("s_sub_u32 %s, %s, %s" % (R("s", self.s_par), R("s", self.s_tile), R("s", t0)))
            e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_par), c.TPI - 1))
            e("s_cselect_b32 %s, 2, 1" % R("s", t1))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_par))
            e("s_cselect_b32 %s, 0, %s" % (R("s", t1), R("s", t1)))
            e("s_lshl_b32 %s, %s, 2" % (R("s", t0), R("s", t1)))
            e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", self.s_w)))
            e("s_lshl_b32 %s, %s, 8" % (R("s", t0), R("s", t0)))
            e("v_lshl_add_u32 %s, %s, 2, %s" % (R("v", v[3]), R("v", lane), R("s", t0)))
            e("buffer_load_dword %s, %s, %s, 0 offen" % (R("v", self.v_tab), R("v", v[3]), R("s", self.srdK, 4)))
            if c.bnin:   # the transform table: behind the piece tables
                e("buffer_load_dword %s, %s, %s, 0 offen offset:%d" % (R("v", self.v_tab2), R("v", v[3]), R("s", self.srdK, 4), NCLS * 1024))
        lgn = c.WN.bit_length() - 1
        e("s_lshr_b32 %s, %s, %d" % (R("s", self.s_wm), R("s", self.s_w), lgn))
        e("s_and_b32 %s, %s, %d" % (R("s", self.s_wn), R("s", self.s_w), c.WN - 1))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_ldsBw), R("s", self.s_w), c.BSTAGE // 4), "this wave's quarter of a weight stage")
        # ---- DMA lane parts first: the first loads wait for nothing else
        # A: x'' = xb*8 + (lane >> 3); j = ((lane & 7) - ((lane >> 3) & 6)) & 7
        l3, l7, j, x, off = v[3], v[4], v[5], v[6], v[7]
        e("v_lshrrev_b32 %s, 3, %s" % (R("v", l3), R("v", lane)))
        e("v_and_b32 %s, 7, %s" % (R("v", l7), R("v", lane)))
        e("v_and_b32 %s, 6, %s" % (R("v", j), R("v", l3)))
        e("v_sub_u32 %s, %s, %s" % (R("v", j), R("v", l7), R("v", j)))
        e("v_and_b32 %s, 7, %s" % (R("v", j), R("v", j)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", j), R("v", j)))
        if c.bnin:
            e("v_lshlrev_b32 %s, 1, %s" % (R("v", self.v_ssoff), R("v", j)), "this lane's 16 bytes of a pixel's chunk = 8 channels = 32 bytes of scale / shift")
            e("v_lshlrev_b32 %s, 4, %s" % (R("v", self.v_lane16), R("v", lane)))
        for xb in range(c.P // 8):
            if xb * 8 > c.W:
                continue
            e("v_add_u32 %s, %d, %s" % (R("v", x), xb * 8 - 1, R("v", l3)), "input x of this lane's position")
            e("v_mov_b32 %s, %d" % (R("v", off), c.Cin * c.ES))
            e("v_mad_u32_u24 %s, %s, %s, %s" % (R("v", off), R("v", x), R("v", off), R("v", j)))
            e("v_cmp_gt_u32 vcc, %d, %s" % (c.W, R("v", x)), "0 <= x < W (x = -1 wraps to 2^32 - 1)")
            e("v_mov_b32 %s, 0x80000000" % R("v", x))
            e("v_cndmask_b32 %s, %s, %s, vcc" % (R("v", self.vA_dma[xb]), R("v", x), R("v", off)))
        # B (variant ib = piece & 1): rr = 8*ib + (lane>>3); channel = (2*ib + (lane>>5))*8 + ((lane>>3)&3);
        #   chunk = (lane&7) ^ (4*ib + ((lane>>4)&3))
        l5, l43, ch, x = v[5], v[8], v[7], v[9]
        e("v_lshrrev_b32 %s, 5, %s" % (R("v", l5), R("v", lane)))
        e("v_bfe_u32 %s, %s, 4, 2" % (R("v", l43), R("v", lane)))
        for ib in range(2):
            e("v_lshl_add_u32 %s, %s, 3, %d" % (R("v", ch), R("v", l5), 16 * ib))
            e("v_and_b32 %s, 3, %s" % (R("v", x), R("v", l3)))
            e("v_add_u32 %s, %s, %s" % (R("v", ch), R("v", ch), R("v", x)))
            e("v_mov_b32 %s, %d" % (R("v", x), c.w_row))
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", ch), R("v", ch), R("v", x)))
            e("v_or_b32 %s, %d, %s" % (R("v", x), 4 * ib, R("v", l43)))
            e("v_xor_b32 %s, %s, %s" % (R("v", x), R("v", l7), R("v", x)))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", self.vB_dma[ib]), R("v", x), R("v", ch)))

        # ---- descriptors ------------------------------------------------------------------------------------------
        e("s_waitcnt lgkmcnt(0)")
        if c.fp8:
            # output scale: oscale alone, or oscale / (*scale_in * *scale_wt) with the two per-tensor scales read from device memory (delayed scaling)
            lab = self.newlabel("noscale")
            e("v_mov_b32 %s, %s" % (R("v", self.v_osc), R("s", self.s_x4 + 1)))
            e("s_cmp_eq_u64 %s, 0" % R("s", self.s_sci, 2))
            e("s_cbranch_scc1 %s" % lab)
            e("s_load_dword %s, %s, 0x0" % (R("s", t0), R("s", self.s_sci, 2)))
            e("s_load_dword %s, %s, 0x0" % (R("s", t1), R("s", self.s_x4 + 2, 2)))
            e("s_waitcnt lgkmcnt(0)")
            e("v_mov_b32 %s, %s" % (R("v", self.v_osc), R("s", t0)))
            e("v_mul_f32 %s, %s, %s" % (R("v", self.v_osc), R("s", t1), R("v", self.v_osc)))
            e("v_rcp_f32 %s, %s" % (R("v", self.v_osc), R("v", self.v_osc)))
            e("s_nop 1")
            e("v_mul_f32 %s, %s, %s" % (R("v", self.v_osc), R("s", self.s_x4 + 1), R("v", self.v_osc)))
            self.label(lab)
        self.comment("descriptors: A = this tile's images, B = this column tile's weight rows, O = this tile's output pixels")
        tile_out = c.tile_rows * c.W * c.NCOLS * 2 * (4 if c.s2d else 1)
        rowb = c.W * c.Cin * c.ES
        if c.ROWS_T:
            # A window of a row tile: LROWS image rows starting ONE ROW ABOVE the tile (the first tile of an image never touches
            # that row, the last never the row below: those table slots repeat another piece).  tile*ROWS_T rows - 1 row, 64-bit
            tile_in = c.LROWS * rowb
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_tile), c.ROWS_T * rowb))
            e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_tile), c.ROWS_T * rowb))
            e("s_sub_u32 %s, %s, %d" % (R("s", t0), R("s", t0), rowb))
            e("s_subb_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
        else:
            tile_in = c.IPT * c.H * rowb
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_tile), tile_in))
            e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_tile), tile_in))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdA), R("s", ka + 0), R("s", t0)))
        e("s_addc_u32 %s, %s, %s" % (R("s", self.srdA + 1), R("s", ka + 1), R("s", t1)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdA + 1), R("s", self.srdA + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdA + 2), tile_in))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdA + 3))
        if c.bnin:
            # a out: the same window on the other tensor; its ReLU bits: 1 byte per 16 bytes of it (the window's offset / 16: rowb is a multiple of 16);
            # scale / shift: [2][Cin] floats
            e("s_add_u32 %s, %s, %s" % (R("s", self.srdA1), R("s", ka + 8), R("s", t0)))
            e("s_addc_u32 %s, %s, %s" % (R("s", self.srdA1 + 1), R("s", ka + 9), R("s", t1)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdA1 + 1), R("s", self.srdA1 + 1)))
            e("s_mov_b32 %s, %d" % (R("s", self.srdA1 + 2), tile_in))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdA1 + 3))
            assert rowb % 16 == 0 and tile_in % 16 == 0
            if c.ROWS_T:
                e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_tile), c.ROWS_T * rowb // 16))
                e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_tile), c.ROWS_T * rowb // 16))
                e("s_sub_u32 %s, %s, %d" % (R("s", t0), R("s", t0), rowb // 16))
                e("s_subb_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
            else:
                e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_tile), tile_in // 16))
                e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_tile), tile_in // 16))
            e("s_add_u32 %s, %s, %s" % (R("s", self.srdBt), R("s", ka + 10), R("s", t0)))
            e("s_addc_u32 %s, %s, %s" % (R("s", self.srdBt + 1), R("s", ka + 11), R("s", t1)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdBt + 1), R("s", self.srdBt + 1)))
            e("s_mov_b32 %s, %d" % (R("s", self.srdBt + 2), tile_in // 16))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdBt + 3))
            e("s_mov_b32 %s, %s" % (R("s", self.srdSS), R("s", ka + 12)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdSS + 1), R("s", ka + 13)))
            e("s_mov_b32 %s, %d" % (R("s", self.srdSS + 2), 2 * c.Cin * 4))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdSS + 3))
            e("s_mov_b32 %s, 0" % R("s", self.s_last))
            e("s_mov_b32 %s, 0x00010001" % R("s", self.s_k1))
        # B: rows nt*BN .. + BN
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_nt), c.BN * c.w_row))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdB), R("s", ka + 2), R("s", t0)))
        e("s_addc_u32 %s, %s, 0" % (R("s", self.srdB + 1), R("s", ka + 3)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdB + 1), R("s", self.srdB + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdB + 2), c.BN * c.w_row))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdB + 3))
        e("s_mov_b32 %s, %s" % (R("s", self.s_nch), R("s", kb + 3)))
        e("s_mov_b32 %s, %s" % (R("s", self.s_wt + 8), R("s", kb + 2)))
        # the table (24 words) to its place: [NPA lds][NPA src][8 B consts]
        def first_loads():
            # ---- first loads: weight stages 0 .. NB-1, the A tile of chunk 0 (its pieces never overlap the static zero fill below)
            self.comment("first loads: weight stages, A tile of chunk 0")
            if not self.tab_sgpr:
                e("s_waitcnt vmcnt(0)", "the piece table")
            e("s_mov_b32 %s, 0" % R("s", self.s_cC))
            if c.bnin:
                for ins in self.ss_loads(self.s_cC, tagged=False):
                    e(ins)
            self.first_stage_issue(0)
            if self.dynamic_halo_blocks():
                # rows that are data in one tile parity and zero halo in the other: zeroed by every tile BEFORE its pieces land
                self.zero_blocks([c.ABASE + b * c.ASTRIDE + o for b in range(c.NA) for o in self.dynamic_halo_blocks()])
                e("s_waitcnt lgkmcnt(0)")
                e("s_barrier")
            for i in range(self.NPA):
                for ins in self.a_piece_insts(i, 0, self.s_cC):
                    e(ins)
            for st in range(1, c.NB - (1 if c.fp8 else 0)):   # (fp8: stage t requests stage t + 2 itself)
                self.first_stage_issue(st)


        def descriptors_out():
            # ---- the rest of the set-up runs under the latency of those loads
            # O: + tile*tile_out + nt*512 bytes
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_tile), tile_out))
            e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_tile), tile_out))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_stg), R("s", self.s_nt), c.BN * 2))
            e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", self.s_stg)))
            e("s_addc_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
            e("s_add_u32 %s, %s, %s" % (R("s", self.srdO), R("s", ka + 4), R("s", t0)))
            e("s_addc_u32 %s, %s, %s" % (R("s", self.srdO + 1), R("s", ka + 5), R("s", t1)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdO + 1), R("s", self.srdO + 1)))
            e("s_mov_b32 %s, %d" % (R("s", self.srdO + 2), tile_out))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdO + 3))
            # (the O window starts at this tile's column offset: num_records covers exactly what may be stored)
            e("s_sub_u32 %s, %s, %s" % (R("s", self.srdO + 2), R("s", self.srdO + 2), R("s", self.s_stg)))
            e("s_mov_b32 %s, %s" % (R("s", self.srdX), R("s", ka + 6)), "statistics rows")
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdX + 1), R("s", ka + 7)))
            e("s_mov_b32 %s, 0x7fffffff" % R("s", self.srdX + 2))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdX + 3))
            if c.stats >= 2:
                # y: laid out like the output (same window); mask bytes: 1/16 of it; mean / invstd: this column tile's 256 floats
                e("s_add_u32 %s, %s, %s" % (R("s", self.srdY), R("s", ka + 8), R("s", t0)))
                e("s_addc_u32 %s, %s, %s" % (R("s", self.srdY + 1), R("s", ka + 9), R("s", t1)))
                e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdY + 1), R("s", self.srdY + 1)))
                e("s_mov_b32 %s, %s" % (R("s", self.srdY + 2), R("s", self.srdO + 2)))
                e("s_mov_b32 %s, 0x00020000" % R("s", self.srdY + 3))
                e("s_lshr_b32 %s, %s, 4" % (R("s", t0), R("s", t0)))
                e("s_lshl_b32 %s, %s, 28" % (R("s", self.s_stg), R("s", t1)))
                e("s_or_b32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", self.s_stg)))
                e("s_lshr_b32 %s, %s, 4" % (R("s", t1), R("s", t1)))
                e("s_add_u32 %s, %s, %s" % (R("s", self.srdM), R("s", ka + 10), R("s", t0)))
                e("s_addc_u32 %s, %s, %s" % (R("s", self.srdM + 1), R("s", ka + 11), R("s", t1)))
                e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdM + 1), R("s", self.srdM + 1)))
                e("s_lshr_b32 %s, %s, 4" % (R("s", self.srdM + 2), R("s", self.srdO + 2)))
                e("s_mov_b32 %s, 0x00020000" % R("s", self.srdM + 3))
                e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_nt), c.BN * 4))
                for srd, k0 in ((self.srdMu, 12), (self.srdIs, 14)):
                    e("s_add_u32 %s, %s, %s" % (R("s", srd), R("s", ka + k0), R("s", t0)))
                    e("s_addc_u32 %s, %s, 0" % (R("s", srd + 1), R("s", ka + k0 + 1)))
                    e("s_and_b32 %s, %s, 0xffff" % (R("s", srd + 1), R("s", srd + 1)))
                    e("s_mov_b32 %s, %d" % (R("s", srd + 2), c.BN * 4))
                    e("s_mov_b32 %s, 0x00020000" % R("s", srd + 3))

        def lane_out():
            # ---- output lane offset: pixel part * NCOLS*2 + (wn*NT*16 + kg*8)*2
            x, off = v[6], v[7]
            if c.P >= 16:
                e("v_mov_b32 %s, %s" % (R("v", x), R("v", r)))
            else:  # P == 8: two image rows per fragment
                e("v_lshrrev_b32 %s, 3, %s" % (R("v", x), R("v", r)))
                e("v_mul_u32_u24 %s, %d, %s" % (R("v", x), c.W * (2 if c.s2d else 1), R("v", x)))   # (s2d: 2 output rows of 2W pixels, halved here, doubled below)
                e("v_and_b32 %s, 7, %s" % (R("v", off), R("v", r)))
                e("v_add_u32 %s, %s, %s" % (R("v", x), R("v", x), R("v", off)))
            if c.s2d:
                e("v_lshlrev_b32 %s, 1, %s" % (R("v", x), R("v", x)), "dy position (i, j) -> output pixel (2i, 2j) of the class's plane")
            e("v_mov_b32 %s, %d" % (R("v", off), c.NCOLS * 2))
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", x), R("v", x), R("v", off)))
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 2))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", x), R("v", kg), R("v", x)))
            e("v_add_u32 %s, %s, %s" % (R("v", self.v_out), R("s", t0), R("v", x)))


        if c.stats >= 2:
            # the BN-backward inputs of tile pairs 0 and 1 are requested FIRST (oldest vector-memory operations: every counted wait
            # of the main loop is unaffected), so the epilogue finds them in registers
            descriptors_out()
            lane_out()
            self.tile_mask_loads()
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 4))
            e("v_lshl_add_u32 %s, %s, 5, %s" % (R("v", self.v_chan), R("v", self.v_kg), R("s", t0)), "this lane's 8 floats of mean / invstd")
            self.epi_issue_loads(0)
            if not getattr(self, "late_pair1", False):
                self.epi_issue_loads(1)
            first_loads()
        else:
            first_loads()
            descriptors_out()
            lane_out()
        # ---- A read bases: pos0 = wm*MFR*16 + r + kx ; chunk = (kg + 4kk + (pos0 & 6)) & 7
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wm), c.MFR * 16 * 128))
        for kx in range(3):
            t, ee, cc = v[3], v[4], v[5]
            e("v_add_u32 %s, %d, %s" % (R("v", t), kx, R("v", r)))
            e("v_and_b32 %s, 6, %s" % (R("v", ee), R("v", t)))
            e("v_add_u32 %s, %s, %s" % (R("v", cc), R("v", ee), R("v", kg)))
            e("v_and_b32 %s, 7, %s" % (R("v", cc), R("v", cc)))
            e("v_lshlrev_b32 %s, 4, %s" % (R("v", cc), R("v", cc)))
            e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", cc), R("v", t), R("v", cc)))
            e("v_add_u32 %s, %s, %s" % (R("v", cc), R("s", t0), R("v", cc)))
            e("v_add_u32 %s, %d, %s" % (R("v", self.vA_rd[kx][0]), c.ABASE, R("v", cc)))
            e("v_xor_b32 %s, 64, %s" % (R("v", self.vA_rd[kx][1]), R("v", self.vA_rd[kx][0])))
        # ---- B read bases: row = wn*NT*16 + r ; chunk = (kg + 4kk) ^ ((r >> 1) & 7)
        sw, cc = v[3], v[4]
        e("v_bfe_u32 %s, %s, 1, 3" % (R("v", sw), R("v", r)))
        e("v_xor_b32 %s, %s, %s" % (R("v", cc), R("v", kg), R("v", sw)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", cc), R("v", cc)))
        e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", cc), R("v", r), R("v", cc)))
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 128))
        e("s_add_u32 %s, %s, %d" % (R("s", t0), R("s", t0), c.BBASE))
        e("v_add_u32 %s, %s, %s" % (R("v", self.vB_rd[0][0]), R("s", t0), R("v", cc)))
        e("v_xor_b32 %s, 64, %s" % (R("v", self.vB_rd[0][1]), R("v", self.vB_rd[0][0])))
        for st in range(1, c.NB):
            for kk in range(2):
                e("v_add_u32 %s, %d, %s" % (R("v", self.vB_rd[st][kk]), st * c.BSTAGE, R("v", self.vB_rd[0][kk])))
        # ---- zero the LDS blocks of both A buffers that no DMA piece writes (halo rows, padding): wave w takes blocks w, w+4, ...
        self.comment("zero the never-written blocks of both A buffers")
        self.zero_blocks([c.ABASE + b * c.ASTRIDE + o for b in range(c.NA) for o in self.never_written_blocks()])
        # accumulators = 0
        for i in range(self.nagpr):
            e("v_accvgpr_write_b32 a%d, 0" % i)
        e("s_waitcnt vmcnt(%d)" % (c.NPB * (c.NB - 1 - (1 if c.fp8 else 0))))
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        if c.bnin:
            # chunk 0's tile has landed for every wave: BatchNorm + ReLU of this wave's blocks (the next block's read under the work on this one),
            # published by a second barrier
            self.comment("BatchNorm + ReLU of the staged tile of chunk 0")
            ns = len(self.tr)
            for ins in self.tr_read(0, 0, 0):
                e(ins)
            for k in range(self.NPA):
                e("s_waitcnt lgkmcnt(%d)" % (1 if k else 0), "this block's read (the previous block's ds_write is younger)")
                if k + 1 < self.NPA:
                    for ins in self.tr_read(k + 1, 0, (k + 1) % ns):
                        e(ins)
                for grp in self.tr_work(k, self.s_cC, k % ns, tagged=False):
                    for ins in grp:
                        e(ins)
            e("s_waitcnt lgkmcnt(0)")
            e("s_barrier")
        # fragments of (stage 0, kk 0)
        rt0 = {stg[0][0] for stg in self.class_stages()}
        assert len(rt0) == 1, "every class starts with the same read tap (the first fragment reads are common)"
        if c.fp8:
            for ins in self.a8_reads(0, rt0.pop()) + self.b8_reads(0, 0) + self.b8_reads(1, 0):
                e(ins)
        else:
            for ins in self.frag_reads(0, rt0.pop(), 0, 0):
                e(ins)
        e("s_mov_b32 %s, %s" % (R("s", self.s_cnt), R("s", self.s_nch)))

    def zero_blocks(self, blocks):
        """ds_write zeros over the listed 1 KiB LDS blocks: the four waves take blocks[4k + w]"""
        c, e = self.c, self.e
        if not blocks:
            return
        v, t0 = self.v_t, self.s_t0
        z = self.F[0][0]
        for i in range(4):
            e("v_mov_b32 %s, 0" % R("v", z + i))
        e("v_and_b32 %s, 63, v0" % R("v", v[8]))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", v[8]), R("v", v[8])), "lane*16")
        for k in range((len(blocks) + 3) // 4):
            grp = blocks[4 * k:4 * k + 4]
            while len(grp) < 4:      # a short last group repeats its last block (same zeros)
                grp.append(grp[-1])
            e("s_mov_b32 %s, %d" % (R("s", t0), grp[0]))
            for w in range(1, 4):
                if grp[w] != grp[0]:
                    e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_w), w))
                    e("s_cselect_b32 %s, %d, %s" % (R("s", t0), grp[w], R("s", t0)))
            e("v_add_u32 %s, %s, %s" % (R("v", v[9]), R("s", t0), R("v", v[8])))
            e("ds_write_b128 %s, %s" % (R("v", v[9]), R("v", z, 4)))

    VMTAG = "\t;vm:"   # main-loop bookkeeping (stripped before the text is returned): tags a vector-memory operation for the counted waits

    def a_piece_insts(self, k, buf, s_chunk, tag=""):
        """A piece slot k (table entries k and NPA + k of this wave) of the chunk at byte offset s_chunk into A buffer `buf`"""
        c = self.c
        var = a_slots(c)[k][0]
        tag = self.VMTAG + tag if tag else ""
        if self.tab_sgpr:
            return ["s_add_u32 m0, %s, %d" % (R("s", self.s_tbl + k), c.ABASE + buf * c.ASTRIDE),
                    "s_add_u32 %s, %s, %s" % (R("s", self.s_t1), R("s", s_chunk), R("s", self.s_tbl + self.NPA + k)),
                    "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vA_dma[var]), R("s", self.srdA, 4), R("s", self.s_t1)) + tag]
        return ["v_readlane_b32 %s, %s, %d" % (R("s", self.s_a), R("v", self.v_tab), k),
                "v_readlane_b32 %s, %s, %d" % (R("s", self.s_b), R("v", self.v_tab), self.NPA + k),
                "s_add_u32 m0, %s, %d" % (R("s", self.s_a), c.ABASE + buf * c.ASTRIDE),
                "s_add_u32 %s, %s, %s" % (R("s", self.s_t1), R("s", s_chunk), R("s", self.s_b)),
                "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vA_dma[var]), R("s", self.srdA, 4), R("s", self.s_t1)) + tag]

    def b_piece_insts(self, i, bp, s_stage, tag=""):
        """weight piece i into ring stage bp; s_stage holds (wtap*Cin + chunk*64)*2"""
        c = self.c
        tag = self.VMTAG + tag if tag else ""
        if self.tab_sgpr:
            return ["s_add_u32 m0, %s, %d" % (R("s", self.s_ldsBw), c.BBASE + bp * c.BSTAGE + i * 1024),
                    "s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", s_stage), R("s", self.s_tbl + 2 * self.NPA + i)),
                    "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vB_dma[i & 1]), R("s", self.srdB, 4), R("s", self.s_t0)) + tag]
        return ["v_readlane_b32 %s, %s, %d" % (R("s", self.s_b), R("v", self.v_tab), 2 * self.NPA + i),
                "s_add_u32 m0, %s, %d" % (R("s", self.s_ldsBw), c.BBASE + bp * c.BSTAGE + i * 1024),
                "s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", s_stage), R("s", self.s_b)),
                "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vB_dma[i & 1]), R("s", self.srdB, 4), R("s", self.s_t0)) + tag]

    def patch_waits(self, start):
        """the stage-barrier waits of the main-loop trip emitted since self.out[start]: a wait line carries `@need:<tags>@`; its count = vector-memory
        operations issued after the youngest operation it names (the most recent one of each tag, looking back cyclically through the trip: steady
        state; the first trip only has MORE younger operations in flight, from the prologue, which makes a counted wait stronger, never weaker)"""
        import re
        ops = []      # (line index, tag) of the tagged operations, in issue order; every other buffer_ instruction of the trip must be tagged
        waits = []    # (line index, number of operations issued before it, needed tags)
        for i in range(start, len(self.out)):
            line = self.out[i]
            m = re.search(r";vm:(\S+)", line)
            if m:
                ops.append((i, m.group(1)))
            elif re.match(r"\tbuffer_", line):
                raise AssertionError("untagged vector-memory operation in the main loop: " + line)
            w = re.search(r"@need:([^@]*)@", line)
            if w:
                waits.append((i, len(ops), w.group(1).split(",")))
        n = len(ops)
        for i, before, need in waits:
            best = None
            for tag in need:
                # distance back to the most recent operation with this tag (cyclic)
                for d in range(1, n + 1):
                    if ops[(before - d) % n][1] == tag:
                        best = d if best is None else min(best, d)
                        break
                else:
                    raise AssertionError("no operation tagged %s in the trip" % tag)
            cnt = best - 1
            assert 0 <= cnt <= 63
            self.out[i] = re.sub(r"vmcnt\(@need:[^@]*@\)", "vmcnt(%d)" % cnt, self.out[i])
        for i in range(start, len(self.out)):
            self.out[i] = re.sub(r"\t;vm:\S+", "", self.out[i])

    def first_stage_issue(self, st):
        """prologue: the weight pieces of the workgroup's stage st (< NB) into ring stage st.  One class: tap st of chunk 0.  Several classes
        (Cfg.s2d): the class's tap st % T of chunk st / T, weight slot and chunk offset selected by s_cls (a chunk past the reduction re-loads
        chunk 0: valid memory, never used)"""
        c, e = self.c, self.e
        cls = self.class_stages()
        if len(cls) == 1:
            assert st < len(cls[0])
            self.b_stage_issue_all(cls[0][st][1], st, self.s_cC)
            return
        sel = [(stg[st % len(stg)][1], st // len(stg)) for stg in cls]
        tmp, tmp2 = self.s_cN, self.s_cNN     # (not live before the main loop)
        e("s_mov_b32 %s, %s" % (R("s", self.s_stg), R("s", self.s_wt + sel[0][0])))
        for k in range(1, len(cls)):
            if sel[k][0] != sel[0][0]:
                e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_cls), k))
                e("s_cselect_b32 %s, %s, %s" % (R("s", self.s_stg), R("s", self.s_wt + sel[k][0]), R("s", self.s_stg)))
        if any(d for _, d in sel):
            e("s_mov_b32 %s, %d" % (R("s", tmp), 128 * sel[0][1]))
            for k in range(1, len(cls)):
                if sel[k][1] != sel[0][1]:
                    e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_cls), k))
                    e("s_cselect_b32 %s, %d, %s" % (R("s", tmp), 128 * sel[k][1], R("s", tmp)))
            e("s_lshl_b32 %s, %s, 7" % (R("s", tmp2), R("s", self.s_nch)))
            e("s_cmp_ge_u32 %s, %s" % (R("s", tmp), R("s", tmp2)))
            e("s_cselect_b32 %s, 0, %s" % (R("s", tmp), R("s", tmp)))
            e("s_add_u32 %s, %s, %s" % (R("s", self.s_stg), R("s", self.s_stg), R("s", tmp)))
        for i in range(c.NPB):
            for ins in self.b_piece_insts(i, st, self.s_stg):
                e(ins)

    def b_stage_issue_all(self, tap, bp, s_chunk):
        e = self.e
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_stg), R("s", self.s_wt + tap), R("s", s_chunk)))
        for i in range(self.c.NPB):
            for ins in self.b_piece_insts(i, bp, self.s_stg):
                e(ins)

    def frag_reads(self, fset, tap, kk, bp):
        """ds_read_b128 list of the fragments of (tap, kk) from the CURRENT A bases and weight stage bp into fragment set fset"""
        c = self.c
        ky, kx = divmod(tap, 3)
        fa, fb = self.F[fset]
        out = []
        # weights first: the first MFMAs of a substep need B0 and A0
        order = []
        for n in range(c.NT):
            order.append(("b", n))
            if n < c.MFR:
                order.append(("a", n))
        for m in range(c.NT, c.MFR):
            order.append(("a", m))
        for kind, i in order:
            if kind == "a":
                off = i * 2048 + ky * c.P * 128  # fragment i starts at position 16*i, the tap's row shift is ky*P positions
                out.append("ds_read_b128 %s, %s offset:%d" % (R("v", fa + 4 * i, 4), R("v", self.vA_rd[kx][kk]), off))
            else:
                out.append("ds_read_b128 %s, %s offset:%d" % (R("v", fb + 4 * i, 4), R("v", self.vB_rd[bp][kk]), i * 2048))
        return out

    def mfmas(self, fset):
        c = self.c
        fa, fb = self.F[fset]
        out = []
        for n in range(c.NT):
            for m in range(c.MFR):
                acc = (m * c.NT + n) * 4
                out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (R("a", acc, 4), R("v", fb + 4 * n, 4), R("v", fa + 4 * m, 4), R("a", acc, 4)))
        return out

    def interleave(self, mf, others, first=1):
        """place the instruction groups `others` (lists) between the MFMAs, evenly, starting after MFMA `first`"""
        n, k = len(mf), len(others)
        slots = {}
        if k:
            span = n - first - 1
            for j, grp in enumerate(others):
                pos = first + (j * span) // k
                slots.setdefault(pos, []).extend(grp)
        for i, m in enumerate(mf):
            self.e(m)
            for ins in slots.get(i, []):
                self.e(ins)

    def skew(self):
        c, e = self.c, self.e
        if not c.skew:
            return
        lab = self.newlabel("skew")
        for k in range(1, 4):
            e("s_cmp_lt_u32 %s, %d" % (R("s", self.s_w), k))
            e("s_cbranch_scc1 %s" % lab)
            e("s_nop %d" % (c.skew - 1))
        self.label(lab)

    # ---- kernel classes: a class = the taps one workgroup accumulates over every chunk, as (read tap, weight slot): read tap ky*3 + kx = the LDS
    # shift of the A fragment reads (position p reads p + ky*P + kx), weight slot = index into the kernel argument wtap[9]
    S2D_CLASSES = (((1, 1), ((0, 0), (0, 1), (1, 0), (1, 1))), ((1, 0), ((0, 0), (1, 0))), ((0, 1), ((0, 0), (0, 1))), ((0, 0), ((0, 0),)))

    def class_stages(self):
        """stride 1: ONE class of 9 taps.  Stride-2 data gradient (Cfg.s2d): the four output-parity classes (ph, pw) of
        dx[n][2i + ph][2j + pw] = sum over the class's taps of dy[n][i + dh][j + dw] * w[tap], longest first (class = workgroup id y / column tiles: the
        long classes are dispatched first); the staged dy tile is dconv's own, a shift (dh, dw) in {0, 1}^2 is read tap (dh + 1, dw + 1)."""
        if not getattr(self.c, "s2d", 0):
            return [[(t, t) for t in range(9)]]
        out, slot = [], 0
        for (ph, pw), taps in self.S2D_CLASSES:
            out.append([((dh + 1) * 3 + (dw + 1), slot + i) for i, (dh, dw) in enumerate(taps)])
            slot += len(taps)
        assert slot == 9
        return out

    def a_carriers(self, T):
        """(taps whose first substep carries A pieces of the next chunk, pieces per such tap): every tap but the last, at most 8.  Cfg.bnin: the pieces
        must have landed for EVERY wave early enough for the transform to run between that barrier and the chunk's last one: see tr_plan()"""
        n = max(1, min(T - 1, 8))
        if self.c.bnin and self.c.NA == 2:
            n = self.tr_plan(T)[0]
        aps = (self.NPA + n - 1) // n
        return (self.NPA + aps - 1) // aps, aps

    def tr_plan(self, T):
        """Cfg.bnin, multi-chunk kernels: (n_a, tc, pps, substeps).  The next chunk's A pieces are issued in the first substeps of taps 0 .. n_a-1; the
        stage barrier of tap tc = n_a + NB - 2 is the first whose counted wait covers all of them (it waits for the weight group issued behind tap
        n_a's... tap tc+1-NB's barrier, which is younger than every A piece), so from the second substep of tap tc on every wave may read any landed
        block.  A block is read (ds_read_b128) in one substep and worked on in the next: the work substeps are (tc+1, 0) .. (T-1, 0), pps pieces each."""
        c = self.c
        assert T == 9
        for pps in (1, 2, 3):
            for n_a in range(6, 0, -1):
                tc = n_a + c.NB - 2
                subs = [(t, sub) for t in range(tc + 1, T) for sub in (0, 1)][:-1]   # (T-1, 1) is behind the publishing barrier
                if len(subs) * pps >= self.NPA and tc + 1 < T:
                    return n_a, tc, pps, subs
        raise AssertionError("no transform plan")

    def npc(self, T, t):
        """A pieces issued in the first substep of tap t (of T)"""
        if self.c.NA != 2:
            return 0
        ntap, aps = self.a_carriers(T)
        return max(0, min((t + 1) * aps, self.NPA) - t * aps) if t < ntap else 0

    # ---- Cfg.bnin: BatchNorm + ReLU of the staged tile, one 1 KiB block (transform slot k of ttables) at a time ---------------------------------------
    def tr_inv_mask(self, var):
        """64-bit mask of the lanes of a variant-`var` block that hold no pixel (x = var*8 - 1 + (lane >> 3) outside [0, W)): their bytes are zeros and stay"""
        m = 0
        for lane in range(64):
            x = var * 8 - 1 + (lane >> 3)
            if not (0 <= x < self.c.W):
                m |= 1 << lane
        return m

    def tr_read(self, k, buf, i):
        """read slot k's block of A buffer `buf` into register set i"""
        c, r = self.c, self.tr[i]
        return ["v_readlane_b32 %s, %s, %d" % (R("s", self.s_ta), R("v", self.v_tab2), k),
                "s_add_u32 %s, %s, %d" % (R("s", self.s_ta), R("s", self.s_ta), c.ABASE + buf * c.ASTRIDE),
                "v_add_u32 %s, %s, %s" % (R("v", r["ta"]), R("s", self.s_ta), R("v", self.v_lane16)),
                "ds_read_b128 %s, %s" % (R("v", r["d"], 4), R("v", r["ta"]))]

    def tr_work(self, k, s_chunk, i, tagged=True):
        """a = relu(y * scale + shift) of the block in register set i, as groups of at most two instructions: back to LDS, to the `a` tensor
        (the LDS-DMA's own offsets) and its ReLU bits (offsets >> 4).  A TR_SKIP slot, and the lanes without a pixel, write nothing to LDS (EXEC
        around the ds_write: their bytes are zeros and stay) and store nothing (offset out of range)."""
        c, r = self.c, self.tr[i]
        var = a_slots(c)[k][0]
        d, f = r["d"], r["f"]
        inv = self.tr_inv_mask(var)
        tag = (self.VMTAG + "S") if tagged else ""
        g = []
        g.append(["v_readlane_b32 %s, %s, %d" % (R("s", self.s_tb), R("v", self.v_tab2), self.NPA + k),
                  "s_or_b32 %s, %s, %s" % (R("s", self.s_tb), R("s", self.s_tb), R("s", self.s_last))])
        g.append(["s_and_b32 %s, %s, 0x%x" % (R("s", self.s_dead), R("s", self.s_tb), TR_SKIP),
                  "s_cselect_b64 %s, -1, 0" % R("s", self.s_sel, 2)])
        g.append(["s_and_b32 %s, %s, 0x%x" % (R("s", self.s_tb), R("s", self.s_tb), TR_SKIP - 1),
                  "s_add_u32 %s, %s, %s" % (R("s", self.s_tb), R("s", self.s_tb), R("s", s_chunk))])
        if inv & 0xFFFFFFFF:
            g.append(["s_or_b32 %s, %s, 0x%x" % (R("s", self.s_sel), R("s", self.s_sel), inv & 0xFFFFFFFF)])
        if inv >> 32:
            g.append(["s_or_b32 %s, %s, 0x%x" % (R("s", self.s_sel + 1), R("s", self.s_sel + 1), inv >> 32)])
        g.append(["v_or_b32 %s, %s, %s" % (R("v", r["o"]), R("s", self.s_dead), R("v", self.vA_dma[var])),
                  "s_lshr_b32 %s, %s, 4" % (R("s", self.s_tb2), R("s", self.s_tb))])
        for j in range(4):
            g.append(["v_lshlrev_b32 %s, 16, %s" % (R("v", f + 2 * j), R("v", d + j)),
                      "v_and_b32 %s, 0xffff0000, %s" % (R("v", f + 2 * j + 1), R("v", d + j))])
        for j in range(4):
            g.append(["v_fma_f32 %s, %s, %s, %s" % (R("v", f + 2 * j + q), R("v", f + 2 * j + q), R("v", self.v_sc + 2 * j + q), R("v", self.v_sh + 2 * j + q)) for q in range(2)])
        # round to bf16 first, ReLU on the packed pairs as signed 16-bit integers (a negative bf16, -0 included, is a negative int16): the same values as
        # ReLU in fp32 followed by the rounding, in 4 instead of 8 instructions
        for j in range(0, 4, 2):
            g.append(["v_cvt_pk_bf16_f32 %s, %s, %s" % (R("v", f + j + q), R("v", f + 2 * (j + q)), R("v", f + 2 * (j + q) + 1)) for q in range(2)])
        for j in range(0, 4, 2):
            g.append(["v_pk_max_i16 %s, %s, 0" % (R("v", f + j + q), R("v", f + j + q)) for q in range(2)])
        # ReLU bits = (a != 0) of the eight stored values (bn_apply_kernel tests the fp32 value: the same bit unless a positive fp32 value rounds to a
        # bf16 zero, below 2^-134): 0 / 1 per half, gathered to bit 2j + half of one byte
        t = f + 4
        for j in range(0, 4, 2):
            g.append(["v_pk_min_u16 %s, %s, %s" % (R("v", t + j + q), R("v", f + j + q), R("s", self.s_k1)) for q in range(2)])
        g.append(["v_lshl_or_b32 %s, %s, 2, %s" % (R("v", t), R("v", t + 1), R("v", t)), "v_lshl_or_b32 %s, %s, 2, %s" % (R("v", t + 2), R("v", t + 3), R("v", t + 2))])
        g.append(["v_lshl_or_b32 %s, %s, 4, %s" % (R("v", t), R("v", t + 2), R("v", t)), "v_lshrrev_b32 %s, 4, %s" % (R("v", r["o2"]), R("v", r["o"]))])
        g.append(["v_lshrrev_b32 %s, 15, %s" % (R("v", t + 1), R("v", t)), "v_and_b32 %s, 0x55, %s" % (R("v", t), R("v", t))])
        g.append(["v_and_b32 %s, 0xaa, %s" % (R("v", t + 1), R("v", t + 1)), "v_or_b32 %s, %s, %s" % (R("v", r["bits"]), R("v", t + 1), R("v", t))])
        # (one group: nothing else runs under the partial EXEC)
        g.append(["s_andn2_b64 exec, exec, %s" % R("s", self.s_sel, 2), "ds_write_b128 %s, %s" % (R("v", r["ta"]), R("v", f, 4)), "s_mov_b64 exec, -1"])
        g.append(["buffer_store_dwordx4 %s, %s, %s, %s offen" % (R("v", f, 4), R("v", r["o"]), R("s", self.srdA1, 4), R("s", self.s_tb)) + tag])
        g.append(["buffer_store_byte %s, %s, %s, %s offen" % (R("v", r["bits"]), R("v", r["o2"]), R("s", self.srdBt, 4), R("s", self.s_tb2)) + tag])
        return g

    def ss_loads(self, s_chunk, tagged=True):
        """scale / shift of this lane's 8 channels of the chunk at byte offset s_chunk (64 channels = 128 bytes of a pixel = 256 bytes of floats)"""
        tag = (self.VMTAG + "C") if tagged else ""
        out = ["s_lshl_b32 %s, %s, 1" % (R("s", self.s_tb), R("s", s_chunk))]
        for h in range(2):
            out.append("buffer_load_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("v", self.v_sc + 4 * h, 4), R("v", self.v_ssoff), R("s", self.srdSS, 4), R("s", self.s_tb), 16 * h) + tag)
        out.append("s_add_u32 %s, %s, %d" % (R("s", self.s_tb), R("s", self.s_tb), self.c.Cin * 4))
        for h in range(2):
            out.append("buffer_load_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("v", self.v_sh + 4 * h, 4), R("v", self.v_ssoff), R("s", self.srdSS, 4), R("s", self.s_tb), 16 * h) + tag)
        return out

    def mainloop(self):
        c, e = self.c, self.e
        cls = self.class_stages()
        if len(cls) == 1:
            self.mainloop_of(cls[0])
            return
        labs = [self.newlabel("class") for _ in cls]
        l_epi = self.newlabel("epi")
        for k in range(1, len(cls)):
            e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_cls), k))
            e("s_cbranch_scc1 %s" % labs[k])
        for k, st in enumerate(cls):
            self.label(labs[k])
            self.mainloop_of(st)
            if k + 1 < len(cls):
                e("s_branch %s" % l_epi)
        self.label(l_epi)

    # ---- Cfg.fp8 ---------------------------------------------------------------------------------------------------------------------------------------
    def a8_reads(self, aset, rt):
        """the pixel fragments of read tap rt into A set `aset`: two ds_read_b128 per fragment (chunks kg and kg + 4 of the row: this lane's 32 bytes)"""
        c = self.c
        ky, kx = divmod(rt, 3)
        return ["ds_read_b128 %s, %s offset:%d" % (R("v", self.A8[aset] + 8 * m + 4 * kk, 4), R("v", self.vA_rd[kx][kk]), m * 2048 + ky * c.P * 128)
                for m in range(c.MFR) for kk in range(2)]

    def b8_reads(self, n, bp):
        """the weight fragment of column n from ring stage bp into slot n % 4"""
        return ["ds_read_b128 %s, %s offset:%d" % (R("v", self.B8[n % 4] + 4 * kk, 4), R("v", self.vB_rd[bp][kk]), n * 2048) for kk in range(2)]

    def mfma8_col(self, n, aset):
        c = self.c
        return ["v_mfma_f32_16x16x128_f8f6f4 %s, %s, %s, %s" % (R("a", (m * c.NT + n) * 4, 4), R("v", self.B8[n % 4], 8), R("v", self.A8[aset] + 8 * m, 8), R("a", (m * c.NT + n) * 4, 4))
                for m in range(c.MFR)]

    def mainloop_fp8(self, stages):
        """e4m3 operands: a stage = one tap of a 128-channel chunk = NT columns of MFR v_mfma_f32_16x16x128_f8f6f4.  Weight ring of THREE stages: stage t
        requests stage t + 2 (into the slot stage t - 1 released at its barrier) in its first columns; the stage barrier stands in front of column NT - 2,
        behind the last weight-fragment read of the stage, and the next stage's first two weight fragments (and, at a chunk switch, its pixel fragments)
        are read behind it; within a chunk the next tap's pixel fragments are read in the columns before the barrier."""
        c, e = self.c, self.e
        T, NT = len(stages), c.NT
        L = c.NA * T
        assert (L % c.NB == 0 or c.NA == 1) and T % c.NB == 0 and NT >= 4 and (T - 1 + c.NB - 1) // T <= 1
        ntap_a, aps = self.a_carriers(T)
        self.comment("---- main loop (e4m3): chunks (2 per trip) x %d taps x %d columns of %d MFMAs" % (T, NT, c.MFR))
        top, done = self.newlabel("loop"), self.newlabel("done")
        self.label(top)
        body = len(self.out)
        aset = 0
        for cp in range(c.NA):
            if c.NA == 1:
                e("s_mov_b32 %s, 0" % R("s", self.s_cN), "one chunk: the weight stages past the last tap re-load chunk 0 (never used)")
            else:
                e("s_add_u32 %s, %s, 128" % (R("s", self.s_cN), R("s", self.s_cC)))
                e("s_cmp_eq_u32 %s, 1" % R("s", self.s_cnt))
                e("s_cbranch_scc0 %s" % (lab := self.newlabel("notlast")))
                e("s_mov_b32 %s, 0" % R("s", self.s_cN))
                self.label(lab)
            for t in range(T):
                rt, ws = stages[t]
                q = cp * T + t
                bp, bp1, bp2 = q % c.NB, (q + 1) % c.NB, (q + 2) % c.NB
                t1 = (t + 1) % T
                switch = (t == T - 1 and c.NA == 2)           # the next stage reads the other A buffer
                # vector-memory requests of the stage, in its first columns: next chunk's A pieces, then the weight pieces of stage t + 2
                t3 = (t + 2) % T
                s_ch = self.s_cC if t + 2 < T else self.s_cN
                vm = [self.a_piece_insts(k, cp ^ 1, self.s_cN, "A") for k in range(t * aps, min((t + 1) * aps, self.NPA))] if (c.NA == 2 and t < ntap_a) else []
                for i in range(c.NPB):
                    g = self.b_piece_insts(i, bp2, self.s_stg, "B%d" % ((q + 2) % L))
                    if i == 0:
                        g = ["s_add_u32 %s, %s, %s" % (R("s", self.s_stg), R("s", self.s_wt + stages[t3][1]), R("s", s_ch))] + g
                    vm.append(g)
                if c.probe & 1:
                    vm = []
                # the next tap's pixel fragments: before the barrier within a chunk, behind it at a chunk switch
                a_next = [[r] for r in self.a8_reads(aset ^ 1, stages[t1][0])]
                pre = NT - 2
                for n in range(NT):
                    self.comment("chunk parity %d tap %d column %d" % (cp, t, n))
                    e("s_waitcnt lgkmcnt(0)")
                    if n == pre:
                        need = ["B%d" % ((q + 1) % L)] + (["A"] if switch else [])
                        e("s_waitcnt vmcnt(@need:%s@)" % ",".join(need))
                        if not c.probe & 4:
                            e("s_barrier")
                        if switch:
                            d = c.ASTRIDE if cp == 0 else -c.ASTRIDE
                            for kx in range(3):
                                for kk in range(2):
                                    rr = R("v", self.vA_rd[kx][kk])
                                    e("v_add_u32 %s, %d, %s" % (rr, d, rr) if d > 0 else "v_subrev_u32 %s, %d, %s" % (rr, -d, rr))
                    groups = []
                    if n + 2 < NT:
                        groups.append(self.b8_reads(n + 2, bp))
                    else:
                        groups.append(self.b8_reads(n + 2 - NT, bp1))
                    if switch:
                        if n >= pre:
                            k0 = (n - pre) * len(a_next) // 2
                            groups += a_next[k0:k0 + (len(a_next) + 1) // 2] if n == pre else a_next[k0:]
                    elif n < pre:
                        per = -(-len(a_next) // pre)
                        groups += a_next[n * per:(n + 1) * per]
                    if n < pre:
                        per = -(-len(vm) // pre)
                        groups = self.merge(groups, vm[n * per:(n + 1) * per])
                    if c.probe & 2:
                        groups = [g for g in groups if not g[0].startswith("ds_read")]
                    self.interleave(self.mfma8_col(n, aset), groups, first=0)
                aset ^= 1
            if c.NA == 1:
                break
            e("s_mov_b32 %s, %s" % (R("s", self.s_cC), R("s", self.s_cN)))
            e("s_sub_u32 %s, %s, 1" % (R("s", self.s_cnt), R("s", self.s_cnt)))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_cnt))
            if cp == 0:
                e("s_cbranch_scc1 %s" % done)
            else:
                e("s_cbranch_scc0 %s" % top)
        assert c.NA == 1 or aset == 0 or (c.NA * T) % 2 == 0
        self.patch_waits(body)
        self.label(done)

    def mainloop_of(self, stages):
        if self.c.fp8:
            return self.mainloop_fp8(stages)
        c, e = self.c, self.e
        T = len(stages)
        L = c.NA * T                       # stages per trip
        assert L % c.NB == 0 or c.NA == 1, "the ring stage of a tap must not depend on the trip"
        D = (T - 1 + c.NB) // T            # chunks ahead a weight stage is requested
        s_c = [self.s_cC, self.s_cN] + ([self.s_cNN] if D >= 2 else [])
        assert D <= 2
        ntap_a, aps = self.a_carriers(T)
        bn = c.bnin and c.NA == 2
        if bn:
            n_a, tc, pps, subs = self.tr_plan(T)
            work_at = {ts: list(range(i * pps, min((i + 1) * pps, self.NPA))) for i, ts in enumerate(subs)}   # (tap, substep) -> slots worked on
            prev = {(t, sub): ((t, 0) if sub else (t - 1, 1)) for t in range(T) for sub in (0, 1)}
            read_at = {prev[ts]: ks for ts, ks in work_at.items()}                                                # ... -> slots read (one substep earlier)
            assert min(read_at) >= (tc, 1)
        self.comment("---- main loop: chunks (2 per trip: the A buffer and the weight-stage parity alternate) x %d taps x 2 substeps" % T)
        top, done = self.newlabel("loop"), self.newlabel("done")
        self.label(top)
        body = len(self.out)
        for cp in range(c.NA):
            if c.NA == 1:
                e("s_mov_b32 %s, 0" % R("s", self.s_cN), "one chunk: the weight stages past the last tap re-load chunk 0 (never used)")
            else:
                # chunk offsets: current and next (the last chunk re-loads chunk 0: valid memory, never used)
                e("s_add_u32 %s, %s, 128" % (R("s", self.s_cN), R("s", self.s_cC)))
                e("s_cmp_eq_u32 %s, 1" % R("s", self.s_cnt))
                if bn:
                    e("s_cselect_b32 %s, 0x%x, 0" % (R("s", self.s_last), TR_SKIP), "the look-ahead tile behind the last chunk is transformed in place and stored nowhere")
                e("s_cbranch_scc0 %s" % (lab := self.newlabel("notlast")))
                e("s_mov_b32 %s, 0" % R("s", self.s_cN))
                self.label(lab)
                if D >= 2:
                    e("s_add_u32 %s, %s, 256" % (R("s", self.s_cNN), R("s", self.s_cC)))
                    e("s_cmp_lt_u32 %s, 3" % R("s", self.s_cnt))
                    e("s_cselect_b32 %s, 0, %s" % (R("s", self.s_cNN), R("s", self.s_cNN)))
            for t in range(T):
                rt, ws = stages[t]
                q = cp * T + t
                bp = q % c.NB                     # ring stage of (chunk parity, tap): 9 % 3 == 0, so NB = 3 does not depend on cp
                bp1 = (q + 1) % c.NB              # ... of the next stage
                # ---- substep kk = 0: compute on set 0, read (t, kk 1) into set 1, one A piece of the next chunk
                self.comment("chunk parity %d tap %d substep 0" % (cp, t))
                e("s_waitcnt lgkmcnt(0)")
                groups = [[r] for r in self.frag_reads(1, rt, 1, bp)]
                mf = self.mfmas(0)
                pieces = [self.a_piece_insts(k, cp ^ 1, self.s_cN, "A") for k in range(t * aps, min((t + 1) * aps, self.NPA))] if (c.NA == 2 and t < ntap_a) else []
                if bn and t == 0:
                    pieces = [self.ss_loads(self.s_cN)] + pieces   # scale / shift of the next chunk (its transform starts behind tap tc's barrier)
                if c.probe & 1:
                    pieces = []
                if c.probe & 2:
                    groups = []
                groups = self.merge(groups, pieces)
                if bn:
                    groups = self.tr_groups(groups, read_at.get((t, 0), []), work_at.get((t, 0), []), cp ^ 1)
                self.interleave(mf, groups)
                # ---- the stage barrier: stage t+1's weights (and after the last tap the next A tile) have landed for every wave
                self.comment("chunk parity %d tap %d substep 1" % (cp, t))
                # younger than stage t+1's pieces: the weight groups of stages t+2 .. t+NB-1 and the A pieces issued since; at the last tap
                # the A pieces must have landed too (patch_waits counts them)
                need = ["B%d" % ((q + 1) % L)] + (["A"] if (t == T - 1 and c.NA == 2) else [])
                e("s_waitcnt vmcnt(@need:%s@)" % ",".join(need))
                e("s_waitcnt lgkmcnt(0)")
                if not c.probe & 4:
                    e("s_barrier")
                self.skew()
                t2 = (t + 1) % T
                if t == T - 1 and c.NA == 2:  # next chunk: the A bases move to the other buffer
                    d = c.ASTRIDE if cp == 0 else -c.ASTRIDE
                    for kx in range(3):
                        for kk in range(2):
                            rr = R("v", self.vA_rd[kx][kk])
                            e("v_add_u32 %s, %d, %s" % (rr, d, rr) if d > 0 else "v_subrev_u32 %s, %d, %s" % (rr, -d, rr))
                groups = [[r] for r in self.frag_reads(0, stages[t2][0], 0, bp1)]
                # weight stage t+NB into ring stage bp (just released by the barrier)
                t3 = (t + c.NB) % T
                s_ch = s_c[(t + c.NB) // T]
                pieces = []
                for i in range(c.NPB):
                    g = self.b_piece_insts(i, bp, self.s_stg, "B%d" % ((q + c.NB) % L))
                    if i == 0:
                        g = ["s_add_u32 %s, %s, %s" % (R("s", self.s_stg), R("s", self.s_wt + stages[t3][1]), R("s", s_ch))] + g
                    pieces.append(g)
                mf = self.mfmas(1)
                if c.probe & 1:
                    pieces = []
                if c.probe & 2:
                    groups = []
                groups = self.merge(groups, pieces)
                if bn:
                    groups = self.tr_groups(groups, read_at.get((t, 1), []), work_at.get((t, 1), []), cp ^ 1)
                self.interleave(mf, groups)
            if c.NA == 1:
                break
            # next chunk
            e("s_mov_b32 %s, %s" % (R("s", self.s_cC), R("s", self.s_cN)))
            e("s_sub_u32 %s, %s, 1" % (R("s", self.s_cnt), R("s", self.s_cnt)))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_cnt))
            if cp == 0:
                e("s_cbranch_scc1 %s" % done)
            else:
                e("s_cbranch_scc0 %s" % top)
        self.patch_waits(body)
        self.label(done)

    def tr_groups(self, groups, reads, works, buf):
        """the substep's instruction groups with the transform's: the block reads first (their data is waited for at the head of the NEXT substep),
        the work on the blocks read in the previous substep spread through the rest, one block after the other (they share scalar temporaries)"""
        rd = [self.tr_read(k, buf, k % len(self.tr)) for k in reads]
        wk = [g for k in works for g in self.tr_work(k, self.s_cN, k % len(self.tr))]
        return rd + self.merge(groups, wk) if (rd or wk) else groups

    @staticmethod
    def merge(a, b):
        """merge two group lists evenly (b spread through a)"""
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

    # -----------------------------------------------------------------------------------------------------------------
    def frag_out(self, f):
        """(exec mask, output byte offset constant) of global fragment f (0 .. NFRAG-1)"""
        c = self.c
        mask = 0
        for r in range(16):
            po = 16 * f + r
            go, xo = divmod(po, c.P)
            if c.IPT > 1:
                i, y = divmod(go, c.SR)
            else:
                i, y = 0, go
            ok = xo < c.W and y < (c.ROWS_T if c.ROWS_T else c.H) and i < c.IPT
            if ok:
                for q in range(4):
                    mask |= 1 << (16 * q + r)
        # constant: pixel index of lane r = 0 of the fragment minus the lane part at r = 0 (which is 0)
        po = 16 * f
        go, xo = divmod(po, c.P)
        if c.IPT > 1:
            i, y = divmod(go, c.SR)
        else:
            i, y = 0, go
        pix = (i * c.H + y) * c.W + xo
        if c.s2d:   # the class's plane: dy position (y, xo) -> output pixel (2y [+ ph], 2xo [+ pw]) of a 2H x 2W image
            pix = (i * 2 * c.H + 2 * y) * 2 * c.W + 2 * xo
        return mask, pix * c.NCOLS * 2

    def emit_frag(self, m, set_mask=True):
        """s_t0 = output byte offset of THIS wave row's fragment m; with set_mask, EXEC = its valid lanes.  Affine in the wave row
        where the geometry allows, else a compare / select chain over the wave rows."""
        c, e = self.c, self.e
        metas = [self.frag_out(wm * c.MFR + m) for wm in range(c.WM)]
        offs = [x[1] for x in metas]
        masks = [x[0] for x in metas]
        d = offs[1] - offs[0] if c.WM > 1 else 0
        if all(offs[i] - offs[0] == i * d for i in range(c.WM)):
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_wm), d))
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_t0), offs[0]))
        else:
            e("s_mov_b32 %s, %d" % (R("s", self.s_t0), offs[0]))
            for wm in range(1, c.WM):
                e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_wm), wm))
                e("s_cselect_b32 %s, %d, %s" % (R("s", self.s_t0), offs[wm], R("s", self.s_t0)))
        if c.s2d:
            e("s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", self.s_t0), R("s", self.s_clsoff)))
        if not set_mask:
            return
        if len(set(masks)) == 1:
            self.set_exec(masks[0])
        else:
            e("s_mov_b32 exec_lo, 0x%x" % (masks[0] & 0xFFFFFFFF))
            e("s_mov_b32 exec_hi, 0x%x" % (masks[0] >> 32))
            for wm in range(1, c.WM):
                if masks[wm] != masks[0]:
                    e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_wm), wm))
                    e("s_cselect_b32 exec_lo, 0x%x, exec_lo" % (masks[wm] & 0xFFFFFFFF))
                    e("s_cselect_b32 exec_hi, 0x%x, exec_hi" % (masks[wm] >> 32))

    # ---- ReLU mask bytes of the BN-backward sums: the whole tile's bytes by one load per 64 positions (lane = position: the NT*2 bytes of
    # this wave's columns), handed to the lanes that need them by ds_bpermute — a byte load per lane and (fragment, tile pair) costs as
    # many cache-line requests as a 16-byte load (profiles/r05_po_probe_matrix.txt: 10-13 us per launch)
    def alloc_tile_masks(self):
        c, V = self.c, self.V
        self.MKG = (c.MFR + 3) // 4                                  # groups of 4 fragments = 64 positions
        nd = c.NT // 2                                               # dwords per position: 4 bytes (4 lane groups) per tile pair
        self.mk = [V.get(nd, min(nd, 4)) for j in range(self.MKG)]
        self.v_mkoff = V.get()
        self.v_bp = V.get()
        self.v_kg8 = V.get()
        self.v_mb = V.get()

    def tile_mask_loads(self):
        """prologue (EXEC all ones): lane l of group j holds position 64 j + l of this wave row = fragment 4 j + (l >> 4), row r = l & 15"""
        c, e = self.c, self.e
        if c.probe & 8:
            return
        v = self.v_t
        l4, r, off, t = v[3], v[1], self.v_mkoff, v[4]
        nd = c.NT // 2
        op = {1: "buffer_load_dword", 2: "buffer_load_dwordx2", 4: "buffer_load_dwordx4"}[nd]
        rowb = c.NCOLS // 8                                          # mask bytes of a pixel
        e("v_lshrrev_b32 %s, 4, %s" % (R("v", l4), R("v", v[0])), "lane >> 4 (v_t[0] = lane)")
        e("v_lshlrev_b32 %s, 2, %s" % (R("v", self.v_bp), R("v", r)))
        e("v_lshlrev_b32 %s, 3, %s" % (R("v", self.v_kg8), R("v", self.v_kg)))
        # pixel part of lane r inside a fragment, in mask bytes
        s2 = 2 if c.s2d else 1   # (s2d: dy position (i, j) -> output pixel (2i, 2j) of a 2W-wide image, + the class's first pixel)
        if c.P >= 16:
            e("v_mul_u32_u24 %s, %d, %s" % (R("v", t), rowb * s2, R("v", r)))
        else:   # P == 8: two image rows per fragment
            e("v_lshrrev_b32 %s, 3, %s" % (R("v", t), R("v", r)))
            e("v_mul_u32_u24 %s, %d, %s" % (R("v", t), c.W * s2, R("v", t)))
            e("v_and_b32 %s, 7, %s" % (R("v", off), R("v", r)))
            e("v_add_u32 %s, %s, %s" % (R("v", t), R("v", t), R("v", off)))
            e("v_mul_u32_u24 %s, %d, %s" % (R("v", t), rowb * s2, R("v", t)))
        if c.s2d:
            e("s_lshr_b32 %s, %s, 4" % (R("s", self.s_t1), R("s", self.s_clsoff)))
            e("v_add_u32 %s, %s, %s" % (R("v", t), R("s", self.s_t1), R("v", t)))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t1), R("s", self.s_wn), c.NT * 2))
        e("v_add_u32 %s, %s, %s" % (R("v", t), R("s", self.s_t1), R("v", t)), "+ this wave's first mask byte of a pixel")
        for j in range(self.MKG):
            # per (wave row, 16-lane group k): the fragment's first pixel in mask bytes, and which of its 16 rows are pixels
            consts, masks = [], []
            for wm in range(c.WM):
                cw, mw = [], 0
                for k in range(4):
                    f = 4 * j + k
                    if f < c.MFR:
                        fm, fo = self.frag_out(wm * c.MFR + f)
                        cw.append(fo // (c.NCOLS * 2) * rowb)
                        mw |= (fm & 0xFFFF) << (16 * k)
                    else:
                        cw.append(0)
                consts.append(cw)
                masks.append(mw)
            e("v_mov_b32 %s, 0x80000000" % R("v", off), "positions that are no pixel: out of range (the load returns zeros)")
            for k in range(4):
                e("s_mov_b32 %s, %d" % (R("s", self.s_t0), consts[0][k]))
                for wm in range(1, c.WM):
                    if consts[wm][k] != consts[0][k]:
                        e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_wm), wm))
                        e("s_cselect_b32 %s, %d, %s" % (R("s", self.s_t0), consts[wm][k], R("s", self.s_t0)))
                e("s_mov_b32 exec_lo, 0x%x" % ((masks[0] >> (16 * k) & 0xFFFF) << (16 * k) & 0xFFFFFFFF))
                e("s_mov_b32 exec_hi, 0x%x" % (((masks[0] >> (16 * k) & 0xFFFF) << (16 * k)) >> 32))
                for wm in range(1, c.WM):
                    if masks[wm] != masks[0]:
                        mk = (masks[wm] >> (16 * k) & 0xFFFF) << (16 * k)
                        e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_wm), wm))
                        e("s_cselect_b32 exec_lo, 0x%x, exec_lo" % (mk & 0xFFFFFFFF))
                        e("s_cselect_b32 exec_hi, 0x%x, exec_hi" % (mk >> 32))
                e("v_add_u32 %s, %s, %s" % (R("v", off), R("s", self.s_t0), R("v", t)))
            e("s_mov_b64 exec, -1")
            e("%s %s, %s, %s, 0 offen" % (op, R("v", self.mk[j], nd), R("v", off), R("s", self.srdM, 4)))

    def tile_mask_fetch(self, m, p):
        """v_mb = the mask dword of (fragment m, tile pair p), shifted so that bit k of its low byte is this lane's element k (EXEC all ones)"""
        return ["s_mov_b64 exec, -1",
                "ds_bpermute_b32 %s, %s, %s offset:%d" % (R("v", self.v_mb), R("v", self.v_bp), R("v", self.mk[m // 4] + p), 64 * (m % 4)),
                "s_waitcnt lgkmcnt(0)",
                "v_lshrrev_b32 %s, %s, %s" % (R("v", self.v_mb), R("v", self.v_kg8), R("v", self.v_mb))]

    def epi_issue_loads(self, p):
        """BN-backward inputs of tile pair p into register set p & 1"""
        c, e = self.c, self.e
        k = p & 1
        if c.probe & 8:
            for i in range(c.MFR + 4):
                e("s_nop 0")
            return
        for m in range(c.MFR):
            self.emit_frag(m, set_mask=False)
            e("buffer_load_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("v", self.ysets[k][m], 4), R("v", self.v_out), R("s", self.srdY, 4), R("s", self.s_t0), p * 64))
        for h in range(2):
            e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", self.msets[k] + 4 * h, 4), R("v", self.v_chan), R("s", self.srdMu, 4), p * 128 + 16 * h))
            e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", self.msets[k] + 8 + 4 * h, 4), R("v", self.v_chan), R("s", self.srdIs, 4), p * 128 + 16 * h))

    def epilogue(self):
        c, e = self.c, self.e
        self.comment("---- epilogue")
        V = self.V
        base = self.F[0][0]           # fragment registers are free now
        tv = [base + i for i in range(8)]            # accumulator values
        dsets = [base + 8 + 4 * i for i in range(4)]  # packed bf16 data, rotating
        xr = [base + 24 + i for i in range(8)]
        s1 = [base + 32 + i for i in range(8)]
        s2 = [base + 40 + i for i in range(8)]
        vst = base + 48               # LDS address of this lane's statistics slot
        yv = [base + 49 + i for i in range(2)]
        assert base + 51 <= self.F[1][1] + 4 * c.NT
        npair = c.NT // 2
        if c.stats >= 2:
            ysets, msets = self.ysets, self.msets
            issue_loads = self.epi_issue_loads
            GL = c.MFR + 4
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier", "every LDS-DMA of the (unused) lookahead has landed: the ring is free for the statistics scratch")
        e("s_nop 15")
        e("s_nop 15")
        late1 = c.stats >= 2 and getattr(self, "late_pair1", False)  # (pk_gen.py: the second register set lives in the fragment registers)
        if late1:
            issue_loads(1)
        if c.stats:
            # STAT scratch: [wm][256 channels][2] floats at LDS 0 ; this lane (r == 15) owns channels wn*128 + p*32 + kg*8 + e
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_wm), c.BN * 8))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t1), R("s", self.s_wn), c.NT * 16 * 8))
            e("s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", self.s_t0), R("s", self.s_t1)))
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_t0), c.BBASE), "the statistics scratch reuses the weight ring")
            e("v_lshl_add_u32 %s, %s, 6, %s" % (R("v", vst), R("v", self.v_kg), R("s", self.s_t0)))
        for p in range(npair):
            if c.stats:
                for i in range(8):
                    e("v_mov_b32 %s, 0" % R("v", s1[i]))
                    e("v_mov_b32 %s, 0" % R("v", s2[i]))
            if c.stats >= 2 and (p >= 2 or (late1 and p == 1)):
                # pair p's loads were issued behind pair p - 2; younger: pair p - 1's stores (+ pair p + 1's loads)
                e("s_waitcnt vmcnt(%d)" % (0 if c.probe & 40 else c.MFR + (GL if p + 1 < npair else 0)))
            for m in range(c.MFR):
                if c.stats >= 2 and not (c.probe & 24):
                    for ins in self.tile_mask_fetch(m, p):
                        e(ins)
                self.emit_frag(m)
                d = dsets[m % 4]
                for i in range(4):
                    e("v_accvgpr_read_b32 %s, a%d" % (R("v", tv[i]), (m * c.NT + 2 * p) * 4 + i))
                    e("v_accvgpr_read_b32 %s, a%d" % (R("v", tv[4 + i]), (m * c.NT + 2 * p + 1) * 4 + i))
                if c.fp8:
                    for i in range(8):
                        e("v_mul_f32 %s, %s, %s" % (R("v", tv[i]), R("v", self.v_osc), R("v", tv[i])))
                for i in range(4):
                    e("v_cvt_pk_bf16_f32 %s, %s, %s" % (R("v", d + i), R("v", tv[2 * i]), R("v", tv[2 * i + 1])))
                if not (c.probe & 32):
                    e("buffer_store_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("v", d, 4), R("v", self.v_out), R("s", self.srdO, 4), R("s", self.s_t0), p * 64))
                else:
                    e("s_nop 0")
                if c.stats and not (c.probe & 16):
                    for i in range(4):
                        e("v_lshlrev_b32 %s, 16, %s" % (R("v", xr[2 * i]), R("v", d + i)))
                        e("v_and_b32 %s, 0xffff0000, %s" % (R("v", xr[2 * i + 1]), R("v", d + i)))
                if c.stats == 1 and not (c.probe & 16):
                    for i in range(8):
                        e("v_add_f32 %s, %s, %s" % (R("v", s1[i]), R("v", s1[i]), R("v", xr[i])))
                        e("v_fma_f32 %s, %s, %s, %s" % (R("v", s2[i]), R("v", xr[i]), R("v", xr[i]), R("v", s2[i])))
                if c.stats >= 2 and not (c.probe & 16):
                    yr, br = ysets[p & 1][m], self.v_mb
                    for i in range(8):
                        t = tv[i]  # (the accumulator copies are dead after the conversion)
                        e("v_bfe_i32 %s, %s, %d, 1" % (R("v", t), R("v", br), i), "0 / -1: ReLU mask bit of element %d" % i)
                        if c.stats == 3:   # leaky: dz = bit ? dx : dx * 0.01 (the fp32 product of the rounded value, as bn_reduce_kernel<MASK = 3>)
                            t2 = tv[(i + 1) % 8]
                            e("v_mul_f32 %s, 0x%08x, %s" % (R("v", t2), LEAKY_BITS, R("v", xr[i])))
                            e("v_bfi_b32 %s, %s, %s, %s" % (R("v", xr[i]), R("v", t), R("v", xr[i]), R("v", t2)), "dz")
                        else:
                            e("v_and_b32 %s, %s, %s" % (R("v", xr[i]), R("v", xr[i]), R("v", t)), "dz")
                        if i & 1:
                            e("v_and_b32 %s, 0xffff0000, %s" % (R("v", yv[1]), R("v", yr + i // 2)))
                        else:
                            e("v_lshlrev_b32 %s, 16, %s" % (R("v", yv[0]), R("v", yr + i // 2)))
                        e("v_add_f32 %s, %s, %s" % (R("v", s1[i]), R("v", s1[i]), R("v", xr[i])))
                        e("v_fma_f32 %s, %s, %s, %s" % (R("v", s2[i]), R("v", xr[i]), R("v", yv[i & 1]), R("v", s2[i])), "sum dz*y")
            if c.stats:
                e("s_mov_b64 exec, -1")
                e("s_nop 1")
                for sh in (1, 2, 4, 8):
                    for arr in (s1, s2):
                        for i in range(8):
                            rr = R("v", arr[i])
                            e("v_add_f32_dpp %s, %s, %s row_shr:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (rr, rr, rr, sh))
                if c.stats >= 2:
                    # sum dz*xhat = invstd * (sum dz*y - mean * sum dz)
                    mu, isd = msets[p & 1], msets[p & 1] + 8
                    for i in range(8):
                        e("v_mul_f32 %s, %s, %s" % (R("v", tv[i]), R("v", mu + i), R("v", s1[i])))
                    for i in range(8):
                        e("v_sub_f32 %s, %s, %s" % (R("v", s2[i]), R("v", s2[i]), R("v", tv[i])))
                    for i in range(8):
                        e("v_mul_f32 %s, %s, %s" % (R("v", s2[i]), R("v", isd + i), R("v", s2[i])))
                # lanes 15 of every row write their 8 channel slots: [channel][2]
                self.set_exec(0x8000800080008000)
                for i in range(8):
                    e("ds_write_b32 %s, %s offset:%d" % (R("v", vst), R("v", s1[i]), p * 32 * 8 + i * 8))
                    e("ds_write_b32 %s, %s offset:%d" % (R("v", vst), R("v", s2[i]), p * 32 * 8 + i * 8 + 4))
                e("s_mov_b64 exec, -1")
                if c.stats >= 2 and p + 2 < npair:
                    issue_loads(p + 2)
        if c.stats:
            # partial row of this workgroup: row[c] = sum, row[NCOLS + c] = sum of squares, c = nt*256 + tid
            e("s_waitcnt lgkmcnt(0)")
            e("s_barrier")
            a0, a1, b0, b1, ad, go = tv[0], tv[1], tv[2], tv[3], tv[4], tv[5]
            if c.BN < 256:
                e("v_cmp_gt_u32 vcc, %d, v0" % c.BN)
                e("s_nop 4")
                e("s_and_b64 exec, exec, vcc", "one thread per channel of the column tile")
            e("v_lshlrev_b32 %s, 3, v0" % R("v", ad))
            e("v_add_u32 %s, %d, %s" % (R("v", ad), c.BBASE, R("v", ad)))
            e("ds_read_b64 %s, %s" % (R("v", a0, 2), R("v", ad)))
            # global offset: (tile*2*NCOLS + nt*BN + tid)*4
            if c.s2d:   # one partial row per (tile, class)
                e("s_lshl_b32 %s, %s, 2" % (R("s", self.s_t0), R("s", self.s_tile)))
                e("s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", self.s_t0), R("s", self.s_cls)))
                e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_t0), 2 * c.NCOLS * 4))
            else:
                e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_tile), 2 * c.NCOLS * 4))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t1), R("s", self.s_nt), c.BN * 4))
            e("s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", self.s_t0), R("s", self.s_t1)))
            e("v_lshlrev_b32 %s, 2, v0" % R("v", go))
            for wmi in range(1, c.WM):   # + the other wave rows, in order
                e("ds_read_b64 %s, %s offset:%d" % (R("v", b0, 2), R("v", ad), wmi * c.BN * 8))
                e("s_waitcnt lgkmcnt(0)")
                e("v_add_f32 %s, %s, %s" % (R("v", a0), R("v", a0), R("v", b0)))
                e("v_add_f32 %s, %s, %s" % (R("v", a1), R("v", a1), R("v", b1)))
            e("s_waitcnt lgkmcnt(0)")
            e("buffer_store_dword %s, %s, %s, %s offen" % (R("v", a0), R("v", go), R("s", self.srdX, 4), R("s", self.s_t0)))
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_t1), R("s", self.s_t0), c.NCOLS * 4))
            e("buffer_store_dword %s, %s, %s, %s offen" % (R("v", a1), R("v", go), R("s", self.srdX, 4), R("s", self.s_t1)))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")

    def set_exec(self, mask):
        e = self.e
        lo, hi = mask & 0xFFFFFFFF, mask >> 32
        e("s_mov_b32 exec_lo, 0x%x" % lo)
        e("s_mov_b32 exec_hi, 0x%x" % hi)

    # -----------------------------------------------------------------------------------------------------------------
    def finish(self):
        c = self.c
        name = c.name
        lds = c.NA * c.ABUF + c.NB * c.BSTAGE  # the A buffer(s), the weight ring
        assert lds <= 160 * 1024
        total_v = self.accum_offset + self.nagpr
        self.ka_size = self.KA["size"] + (NCLS * 4 * 256 if getattr(c, "bnin", 0) else 0)   # (bnin: + the transform tables)
        hdr = []
        hdr.append('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        hdr.append("\t.amdhsa_code_object_version 6")
        hdr.append("\t.text")
        hdr.append("\t.protected\t%s" % name)
        hdr.append("\t.globl\t%s" % name)
        hdr.append("\t.p2align\t8")
        hdr.append("\t.type\t%s,@function" % name)
        hdr.append("%s:" % name)
        tail = []
        tail.append("\t.section\t.rodata,\"a\",@progbits")
        tail.append("\t.p2align\t6, 0x0")
        tail.append("\t.amdhsa_kernel %s" % name)
        kd = dict(group_segment_fixed_size=lds, private_segment_fixed_size=0, kernarg_size=self.ka_size,
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
        tail.append("\t.end_amdhsa_kernel")
        tail.append("\t.text")
        tail.append("\t.amdgpu_metadata")
        tail.append("---")
        tail.append("amdhsa.kernels:")
        tail.append("  - .agpr_count:     %d" % self.nagpr)
        tail.append("    .args:")
        off = 0
        for i in range(9):
            tail.append("      - .address_space:  global\n        .offset:         %d\n        .size:           8\n        .value_kind:     global_buffer" % off)
            off += 8
        for i in range(10):
            tail.append("      - .offset:         %d\n        .size:           4\n        .value_kind:     by_value" % off)
            off += 4
        tail.append("      - .offset:         %d\n        .size:           %d\n        .value_kind:     by_value" % (off, self.ka_size - off))
        off = self.ka_size
        assert off == self.ka_size
        tail.append("    .group_segment_fixed_size: %d" % lds)
        tail.append("    .kernarg_segment_align: 8")
        tail.append("    .kernarg_segment_size: %d" % self.ka_size)
        tail.append("    .max_flat_workgroup_size: 256")
        tail.append("    .name:           %s" % name)
        tail.append("    .private_segment_fixed_size: 0")
        tail.append("    .sgpr_count:     %d" % (self.S.n + 6))
        tail.append("    .sgpr_spill_count: 0")
        tail.append("    .symbol:         %s.kd" % name)
        tail.append("    .uniform_work_group_size: 1")
        tail.append("    .uses_dynamic_stack: false")
        tail.append("    .vgpr_count:     %d" % total_v)
        tail.append("    .vgpr_spill_count: 0")
        tail.append("    .wavefront_size: 64")
        tail.append("amdhsa.target:   amdgcn-amd-amdhsa--gfx950")
        tail.append("amdhsa.version:\n  - 1\n  - 2")
        tail.append("...")
        tail.append("\t.end_amdgpu_metadata")
        body = self.out + ["\t.p2align 8", ".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name)]
        self.lds_bytes = lds
        return "\n".join(hdr + body + tail) + "\n"


# ---------------------------------------------------------------------------------------------------------------------
VARIANTS = {
    # name: geometry of the launches it serves (ResNet-50 at 224 px: layer 3 = 14 x 14 x 256, layer 4 = 7 x 7 x 512)
    "dconv_l3_s1": Cfg("dconv_l3_s1", H=14, W=14, P=16, IPT=1, Cin=256, NCOLS=256, stats=1),
    "dconv_l3_s0": Cfg("dconv_l3_s0", H=14, W=14, P=16, IPT=1, Cin=256, NCOLS=256, stats=0),
    "dconv_l3_s2": Cfg("dconv_l3_s2", H=14, W=14, P=16, IPT=1, Cin=256, NCOLS=256, stats=2),
    "dconv_l4_s0": Cfg("dconv_l4_s0", H=7, W=7, P=8, IPT=2, Cin=512, NCOLS=512, stats=0, NB=3),
    "dconv_l4_s1": Cfg("dconv_l4_s1", H=7, W=7, P=8, IPT=2, Cin=512, NCOLS=512, stats=1, NB=3),
    "dconv_l4_s2": Cfg("dconv_l4_s2", H=7, W=7, P=8, IPT=2, Cin=512, NCOLS=512, stats=2, NB=3),
    # layer 2 (28 x 28 x 128 -> 128): a tile = 14 output rows of one image (2 tiles per image), waves 4 (pixels) x 1, 128 columns
    # (4-row tiles — 80 KiB of LDS, two workgroups per CU, what helps the single-chunk 112 x 112 kernels — measured here: forward 60.4 -> 67.2 us, data gradient
    # 77.8 -> 75.1, step 17.76 -> 17.79: seven times the tiles re-stream the 295 KiB of weights of two chunks; not kept)
    "dconv_l2_s0": Cfg("dconv_l2_s0", H=28, W=28, P=32, IPT=1, Cin=128, NCOLS=128, stats=0, WM=4, WN=1, ROWS_T=14),
    "dconv_l2_s1": Cfg("dconv_l2_s1", H=28, W=28, P=32, IPT=1, Cin=128, NCOLS=128, stats=1, WM=4, WN=1, ROWS_T=14),
    "dconv_l2_s2": Cfg("dconv_l2_s2", H=28, W=28, P=32, IPT=1, Cin=128, NCOLS=128, stats=2, WM=4, WN=1, ROWS_T=14),
    # layer 1 (56 x 56 x 64 -> 64): a tile = 4 output rows (14 per image), one 64-channel chunk, 64 columns: 64 KiB of LDS and
    # <= 256 registers, so TWO workgroups share a CU and cover each other's prologue / epilogue
    "dconv_l1_s0": Cfg("dconv_l1_s0", H=56, W=56, P=64, IPT=1, Cin=64, NCOLS=64, stats=0, WM=4, WN=1, NT=4, ROWS_T=4),
    "dconv_l1_s1": Cfg("dconv_l1_s1", H=56, W=56, P=64, IPT=1, Cin=64, NCOLS=64, stats=1, WM=4, WN=1, NT=4, ROWS_T=4),
    "dconv_l1_s2": Cfg("dconv_l1_s2", H=56, W=56, P=64, IPT=1, Cin=64, NCOLS=64, stats=2, WM=4, WN=1, NT=4, ROWS_T=4),
}


def _bn_in():
    """conv2 of the stride-1 bottlenecks at 224 px with bn1 + ReLU in its operand path (Cfg.bnin): the training forward (BN statistics of its own output)"""
    for tag in ("l1", "l2", "l3", "l4"):
        base = VARIANTS["dconv_%s_s1" % tag]
        name = "dconv_%s_s1_bn" % tag
        VARIANTS[name] = Cfg(**{**base.__dict__, "name": name, "bnin": 1})


_bn_in()


def _fp8():
    """the stride-1 3x3 convolutions of layers 2 - 4 on e4m3 operands (BASELINE configs[4]: "fp8 MFMA convs"; the e4m3 training step of resnet_exec.cpp): forward
    with / without the BN statistics, data gradient with the BN-backward sums.  Layer 2: ONE 128-channel chunk (the tile stays staged for the nine taps)."""
    for tag in ("l2", "l3", "l4"):
        for st in (0, 1, 2):
            base = VARIANTS["dconv_%s_s%d" % (tag, st)]
            name = "dconv_%s_s%d_q" % (tag, st)
            VARIANTS[name] = Cfg(**{**base.__dict__, "name": name, "fp8": 1, "NB": 3})


_fp8()


def _stride2_dgrad():
    """the data gradient of the three 3x3 / stride-2 convolutions of ResNet-50 (conv2 of layer2.0 / layer3.0 / layer4.0; torchvision v1.5 strides in the
    3x3): the stride-1 tile geometry of the layer whose resolution dy has, four output-parity classes per tile (Gen.class_stages), BN-backward sums of
    the block's bn1 in the epilogue as in every other data gradient of the step (_s2), or none (_s0: the per-op entry point)."""
    geo = {
        "l2": dict(H=28, W=28, P=32, IPT=1, Cin=128, NCOLS=128, WM=4, WN=1, ROWS_T=14),
        "l3": dict(H=14, W=14, P=16, IPT=1, Cin=256, NCOLS=256),
        "l4": dict(H=7, W=7, P=8, IPT=2, Cin=512, NCOLS=512),
    }
    for tag, kw in geo.items():
        for st in (0, 2):
            name = "dconv_%s_d2_s%d" % (tag, st)
            VARIANTS[name] = Cfg(name, stats=st, s2d=1, **kw)


_stride2_dgrad()


def _leaky_sums():
    """BASELINE configs[3] (BResNet-50): conv2's data gradient with bn1's backward sums under the leaky-ReLU mask (stats 3) — the stride-1 blocks at ResNet-50's
    own shapes, the striding blocks at their input resolution (v2 .. v4, defined below)"""
    for tag in ("l1", "l2", "l3", "l4"):
        base = VARIANTS["dconv_%s_s2" % tag]
        name = "dconv_%s_s3" % tag
        VARIANTS[name] = Cfg(**{**base.__dict__, "name": name, "stats": 3})


_leaky_sums()


def _other_sizes():
    """the same four layers at the other two sizes of the progressive-resize recipe (BASELINE configs[4]: 160 -> 224 -> 320 px, stage schema
    /root/reference/sota_imagenet/arg_parser.py:63-72): a = 160 px (40 / 20 / 10 / 5), b = 320 px (80 / 40 / 20 / 10).  The row pitch is 8 or a
    multiple of 16 positions (a fragment never straddles two image rows), so these widths pad 17 .. 37 % of a row (12.5 % at 224 px)."""
    geo = {
        "l1a": dict(H=40, W=40, P=48, IPT=1, Cin=64, NCOLS=64, WM=4, WN=1, NT=4, ROWS_T=8),
        "l2a": dict(H=20, W=20, P=32, IPT=1, Cin=128, NCOLS=128, WM=4, WN=1, ROWS_T=10),
        "l3a": dict(H=10, W=10, P=16, IPT=1, Cin=256, NCOLS=256),
        "l4a": dict(H=5, W=5, P=8, IPT=4, Cin=512, NCOLS=512, NB=3),
        "l1b": dict(H=80, W=80, P=96, IPT=1, Cin=64, NCOLS=64, WM=4, WN=1, NT=4, ROWS_T=4),
        "l2b": dict(H=40, W=40, P=48, IPT=1, Cin=128, NCOLS=128, WM=4, WN=1, ROWS_T=8),
        "l3b": dict(H=20, W=20, P=32, IPT=1, Cin=256, NCOLS=256, ROWS_T=5),
        "l4b": dict(H=10, W=10, P=16, IPT=1, Cin=512, NCOLS=512, NB=3),
    }
    for tag, kw in geo.items():
        for st in (0, 1, 2):
            name = "dconv_%s_s%d" % (tag, st)
            VARIANTS[name] = Cfg(name, stats=st, **kw)


_other_sizes()


def _variant_model_sizes():
    """BASELINE configs[3]'s 3x3 launches that ResNet-50 does not have (anti-aliased BResNet-50: the stride of a striding block moves behind
    conv2 into the blur pool, so its conv2 runs at the INPUT resolution — 56 x 56 x 128, 28 x 28 x 256, 14 x 14 x 512 — and the deep stem has two
    3x3 convolutions at 112 x 112 on 32 channels zero-padded to 64; /root/reference/configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51).
    Forward (s1), plain data gradient (s0) and the data gradient with the BN-backward sums under the leaky mask (s3)."""
    geo = {
        "v0": dict(H=112, W=112, P=128, IPT=1, Cin=64, NCOLS=64, WM=4, WN=1, NT=4, ROWS_T=2),   # 2-row tiles: 80 KiB of LDS, TWO workgroups per CU (4-row tiles: one)
        "v2": dict(H=56, W=56, P=64, IPT=1, Cin=128, NCOLS=128, WM=4, WN=1, ROWS_T=4),
        "v3": dict(H=28, W=28, P=32, IPT=1, Cin=256, NCOLS=256, ROWS_T=7),
        "v4": dict(H=14, W=14, P=16, IPT=1, Cin=512, NCOLS=512),
    }
    for tag, kw in geo.items():
        for st in (0, 1, 3):
            name = "dconv_%s_s%d" % (tag, st)
            VARIANTS[name] = Cfg(name, stats=st, **kw)


_variant_model_sizes()


def generate(base, **over):
    c = VARIANTS[base]
    if over:
        c = Cfg(**{**c.__dict__, **over})
    g = Gen(c)
    text = g.gen()
    return c, g, text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="build")
    ap.add_argument("--set", action="append", default=[], help="tuning: override a Cfg field (key=int), with --suffix names the kernel")
    ap.add_argument("--suffix", default="")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    over = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.set}
    for name in (a.names or VARIANTS):
        if a.suffix:
            over["name"] = name + a.suffix
        c, g, text = generate(name, **over)
        name = c.name
        if a.suffix:  # tuning builds: the per-wave table as a raw file for tools/micro/dconv_bench.cpp
            import struct
            with open(os.path.join(a.out, name + ".tbl"), "wb") as f:
                f.write(struct.pack("<768I", *[w for par in tables(c) for row in par for w in row]))
        with open(os.path.join(a.out, name + ".s"), "w") as f:
            f.write(text)
        print("%s: %d lines, %d VGPR + %d AGPR, %d SGPR, LDS %d" % (name, text.count("\n"), g.accum_offset, g.nagpr, g.S.n, g.lds_bytes))


if __name__ == "__main__":
    main()
