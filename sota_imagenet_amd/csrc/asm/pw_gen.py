#!/usr/bin/env python3
"""pw_gen.py — generator of the persistent, unit-pipelined gfx950 (MI355X) kernel for the OUTPUT-HEAVY 1x1 convolutions
(short reduction K into many columns N): ResNet-50's conv3 forward of layer 3 (256 -> 1024).

What it replaces: the cuDNN 1x1 conv forward under `model(data)` of the reference (call form
/root/reference/sota_imagenet/callbacks.py:316), the same IgemmArgs contract as conv_igemm.hip (BN statistics in the
epilogue), selected in launch_igemm().  These launches are HBM-bound (129 MB per launch at batch 256: ~23 us) but ran at
59-66 us because every tile's epilogue (convert, store, statistics) ran with the matrix pipe idle and the next tile's
loads not yet issued.

Structure:
  unit        112 pixels x 256 channels x all K (K = 256: 4 stages of 64 channels).  M = 448 * 112 pixels and 4 column
              tiles at batch 256: 1792 units = exactly 7 per CU.  A persistent workgroup (4 waves = one per SIMD) walks a
              contiguous range of units, column tile fastest, so the 112 x K input tile stays in LDS for 4 units.
  waves       1 (M) x 4 (N): a wave owns all 7 pixel fragments x 64 channels = 28 accumulator tiles of 16 x 16
              (v_mfma_f32_16x16x32_bf16, operands swapped) = 112 AGPRs, and there are TWO accumulator sets: while unit u
              accumulates into one, the epilogue of unit u - 1 (accumulator reads, bf16 conversion, 16-byte stores,
              statistics) is issued BETWEEN the MFMAs of unit u out of the other.
  A operand   K/64 planes [112 rows][128 B] (XOR-swizzled 16-byte chunks), plane c refilled with the next pixel tile right
              after its last read (stage c of the last column tile): no second buffer.
  B operand   weight rows [256][128 B] per (column tile, stage), 3-stage ring at a RUN-TIME ring offset (scalar register),
              so the code is unrolled over the two accumulator sets only; rows permuted so that a lane's accumulators of a
              tile pair are 8 consecutive channels.
  waits       one s_barrier per stage behind a counted vmcnt; the count is derived by the generator from the program order of
              every vector-memory instruction (class Tracker), never by hand.
"""
import argparse
import os
import struct
import sys
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dconv_gen import Alloc, R  # noqa: E402


@dataclass
class PwCfg:
    name: str
    K: int            # input channels (reduction), multiple of 64, K/64 planes must fit LDS (K <= 256)
    N: int            # output channels: a power-of-two multiple of 256
    stats: int        # 0 / 1 (BN statistics of the output)
    MT: int = 7       # pixel fragments per unit
    NB: int = 3
    probe: int = 0    # timing probes (WRONG results): 1 no epilogue work in the loop, 2 no stores, 4 no weight DMA in the loop, 8 no barriers, 16 epilogue = accumulator reads only, 32 epilogue without accumulator reads

    @property
    def NCH(self):
        return self.K // 64

    @property
    def NTN(self):
        return self.N // 256

    @property
    def ROWS(self):
        return self.MT * 16

    @property
    def PLANE(self):
        return self.ROWS * 128

    @property
    def w_row(self):
        return self.K * 2

    @property
    def ABASE(self):
        return self.NB * 32768

    @property
    def SBASE(self):
        return self.ABASE + self.NCH * self.PLANE

    @property
    def LDS(self):
        return self.SBASE + self.NTN * 256 * 8


def b_piece_const(c, w, i):
    """source constant (bytes) of weight piece i of wave w: rows 64w + 8i .. + 8 of the stage image; wave w owns columns 64w..
    (tile n = rows/16 of the wave: pair p = n >> 1; odd tiles hold channels +4)"""
    n = i >> 1
    p, odd = n >> 1, n & 1
    return (64 * w + p * 32 + 4 * odd) * c.w_row


def a_slots(c):
    """A pieces (8 rows each) per wave: list of (variant, [per wave (lds offset inside a plane, source row constant in bytes)]).
    A slot holds pieces of ONE parity (the lane part of the source offset depends on it); a wave without a piece left repeats
    its own previous piece of that parity (same bytes to the same place: harmless)."""
    npieces = c.ROWS // 8
    out = []
    for var in range(2):
        lst = [i for i in range(npieces) if (i & 1) == var]
        prev = None
        for k in range(0, len(lst), 4):
            grp = lst[k:k + 4]
            full = [grp[w] if w < len(grp) else prev[w] for w in range(4)]
            prev = full
            out.append((var, [(i * 1024, i * 8 * c.w_row) for i in full]))
    return out


def tables(c):
    sl = a_slots(c)
    rows = []
    for w in range(4):
        words = [g[w][0] for _, g in sl] + [g[w][1] for _, g in sl] + [b_piece_const(c, w, i) for i in range(8)]
        assert len(words) == 16
        rows.append(words + [0] * 16)
    return rows


class Tracker:
    """program order of the vector-memory instructions of the steady-state loop; vmcnt for a wait = the number of UNCONDITIONAL
    instructions issued after the youngest one the wait must retire (conditional ones only make the real count larger)."""

    def __init__(self):
        self.events = []  # ("op", tag, conditional) | ("wait", placeholder index, [required tags])

    def op(self, tag, conditional=False):
        self.events.append(("op", tag, conditional))

    def wait(self, slot, required):
        self.events.append(("wait", slot, required))

    def resolve(self):
        """events of ONE loop trip; the loop is cyclic: search backwards, wrapping once"""
        n = len(self.events)
        res = {}
        for idx, ev in enumerate(self.events):
            if ev[0] != "wait":
                continue
            _, slot, required = ev
            best = None
            for tag in required:
                cnt, found = 0, False
                for back in range(1, 2 * n + 1):
                    e2 = self.events[(idx - back) % n]
                    if e2[0] != "op":
                        continue
                    if e2[1] == tag:
                        found = True
                        break
                    if not e2[2]:
                        cnt += 1
                assert found, "tag %r never issued" % (tag,)
                best = cnt if best is None else min(best, cnt)
            res[slot] = best if best is not None else 63
        return res


class Gen:
    KA = dict(in_=0, wt=8, out=16, stat=24, units=80, upw=84, mtiles=88, table=128, size=640)

    def __init__(self, c: PwCfg):
        self.c = c
        self.out = []
        self.nlabel = 0
        self.S = Alloc("s", 4, 100)
        self.V = Alloc("v", 1, 256)
        self.tr = Tracker()
        self.wait_slots = {}

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
        c, S, V = self.c, self.S, self.V
        assert c.NTN & (c.NTN - 1) == 0 and c.LDS <= 160 * 1024, c.LDS
        assert c.NCH >= 2, "a plane is refilled one stage before its next read: needs at least two planes"
        self.s_wg = 2
        self.srdA = S.get(4, 4)     # the pixel tile whose planes are being (re)filled
        self.srdB = S.get(4, 4)
        self.srdOp = S.get(4, 4)    # output window of the PREVIOUS unit (null for the first)
        self.srdX = S.get(4, 4)
        self.s_tbl = S.get(16, 4)
        self.s_ka = S.get(16, 4)
        self.s_kb = S.get(4, 4)
        (self.s_u, self.s_uend, self.s_mt, self.s_nt, self.s_ntn, self.s_last, self.s_pnt, self.s_w, self.s_t0, self.s_t1,
         self.s_bcur, self.s_bnext, self.s_fillw, self.s_ldsBw, self.s_bsrc0, self.s_bsrc1, self.s_stg, self.s_mtiles,
         self.s_outlo, self.s_outhi, self.s_alo, self.s_ahi, self.s_t2) = [S.get() for _ in range(23)]
        self.s_bsrc = [self.s_bsrc0, self.s_bsrc1, S.get(), S.get()]   # weight row offset of the column tile of unit u + k
        self.s_tA_lds, self.s_tA_src, self.s_tB = self.s_tbl, self.s_tbl + 4, self.s_tbl + 8

        self.vA_rd = [V.get() for _ in range(2)]
        self.vB0 = [V.get() for _ in range(2)]
        self.vBcur1, self.vBnext0 = V.get(), V.get()
        self.vA_dma = [V.get() for _ in range(2)]
        self.vB_dma = [V.get() for _ in range(2)]
        self.v_out = V.get()
        self.v_st = V.get()
        self.v_kg = V.get()
        self.v_t = [V.get() for _ in range(10)]
        self.F = []
        for s in range(2):
            fa = V.get(4 * c.MT, 4)
            fb = V.get(16, 4)
            self.F.append((fa, fb))
        ep = V.get(48, 4)
        self.tv = [ep + i for i in range(8)]
        self.dsets = [ep + 8 + 4 * i for i in range(4)]
        self.xr = [ep + 24 + i for i in range(8)]
        self.s1 = [ep + 32 + i for i in range(8)]
        self.s2 = [ep + 40 + i for i in range(8)]
        self.nvgpr = V.n
        self.accum_offset = (self.nvgpr + 7) // 8 * 8
        self.nacc = c.MT * 4 * 4          # one accumulator set
        self.nagpr = 2 * self.nacc

        self.UB = 2
        while self.UB * c.NCH < c.NB + 1:
            self.UB += 2
        self.prologue()
        self.loop()
        text = self.finish()
        # resolve the counted waits
        res = self.tr.resolve()
        for slot, n in res.items():
            text = text.replace("@VM%d@" % slot, str(min(n, 63)))
        assert "@VM" not in text
        return text

    # -----------------------------------------------------------------------------------------------------------------
    def a_slot_insts(self, k, plane):
        """LDS-DMA of A slot k into plane `plane` through srdA (the tile set up by the caller)"""
        c = self.c
        var = a_slots(c)[k][0]
        return ["s_add_u32 m0, %s, %d" % (R("s", self.s_tA_lds + k), c.ABASE + plane * c.PLANE),
                "s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_tA_src + k), plane * 128),
                "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vA_dma[var]), R("s", self.srdA, 4), R("s", self.s_t0))]

    def b_piece_insts(self, i, s_dstw, s_src):
        """weight piece i into the ring stage whose per-wave base is s_dstw; s_src = column-tile row offset + chunk*128"""
        return ["s_add_u32 m0, %s, %d" % (R("s", s_dstw), i * 1024),
                "s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", s_src), R("s", self.s_tB + i)),
                "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vB_dma[i & 1]), R("s", self.srdB, 4), R("s", self.s_t0))]

    def set_srdA(self, s_mt_reg):
        """srdA = window of pixel tile s_mt_reg: base + mt*ROWS*K*2, ROWS*K*2 records"""
        c, e = self.c, self.e
        tb = c.ROWS * c.K * 2
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", s_mt_reg), tb))
        e("s_mul_hi_u32 %s, %s, %d" % (R("s", self.s_t1), R("s", s_mt_reg), tb))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdA), R("s", self.s_alo), R("s", self.s_t0)))
        e("s_addc_u32 %s, %s, %s" % (R("s", self.srdA + 1), R("s", self.s_ahi), R("s", self.s_t1)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdA + 1), R("s", self.srdA + 1)))

    def unit_scalars(self):
        """after s_mt / s_nt of the CURRENT unit are set: next column tile, last flag, weight row offsets, refill tile"""
        c, e = self.c, self.e
        e("s_add_u32 %s, %s, 1" % (R("s", self.s_ntn), R("s", self.s_nt)))
        e("s_and_b32 %s, %s, %d" % (R("s", self.s_ntn), R("s", self.s_ntn), c.NTN - 1))
        for k in range(4):
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_nt), k))
            e("s_and_b32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_t0), c.NTN - 1))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_bsrc[k]), R("s", self.s_t0), 256 * c.w_row))
        # refill tile = min(mt + 1, mtiles - 1): its planes are loaded during this unit when it is the last column tile
        e("s_add_u32 %s, %s, 1" % (R("s", self.s_t2), R("s", self.s_mt)))
        e("s_sub_u32 %s, %s, 1" % (R("s", self.s_t0), R("s", self.s_mtiles)))
        e("s_min_u32 %s, %s, %s" % (R("s", self.s_t2), R("s", self.s_t2), R("s", self.s_t0)))
        self.set_srdA(self.s_t2)
        e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_nt), c.NTN - 1))
        e("s_cselect_b32 %s, 1, 0" % R("s", self.s_last))

    def prologue(self):
        c, e = self.c, self.e
        ka, kb, v = self.s_ka, self.s_kb, self.v_t
        t0, t1 = self.s_t0, self.s_t1
        self.comment("---- prologue")
        e("s_load_dwordx8 %s, s[0:1], 0x0" % R("s", ka, 8))          # in, wt, out, stat
        e("s_load_dwordx4 %s, s[0:1], 0x50" % R("s", kb, 4))         # units, units per workgroup, pixel tiles
        lane, r, kg = v[0], v[1], self.v_kg
        e("v_lshrrev_b32 %s, 6, v0" % R("v", v[3]))
        e("v_and_b32 %s, 63, v0" % R("v", lane))
        e("v_readfirstlane_b32 %s, %s" % (R("s", self.s_w), R("v", v[3])))
        e("v_and_b32 %s, 15, v0" % R("v", r))
        e("v_bfe_u32 %s, v0, 4, 2" % R("v", kg))
        e("s_nop 3")
        e("s_lshl_b32 %s, %s, 7" % (R("s", t0), R("s", self.s_w)))
        e("s_add_u32 %s, s0, %s" % (R("s", self.s_t2), R("s", t0)))
        e("s_addc_u32 %s, s1, 0" % R("s", self.s_stg))
        # (s_t2, s_stg) are not an aligned pair: copy into one
        e("s_mov_b32 %s, %s" % (R("s", self.srdX), R("s", self.s_t2)))
        e("s_mov_b32 %s, %s" % (R("s", self.srdX + 1), R("s", self.s_stg)))
        e("s_load_dwordx16 %s, %s, 0x80" % (R("s", self.s_tbl, 16), R("s", self.srdX, 2)))
        e("s_lshl_b32 %s, %s, 13" % (R("s", self.s_ldsBw), R("s", self.s_w)), "this wave's 8 KiB of a weight stage")
        # ---- A DMA lane parts: row = 8i + (lane>>3), chunk = (lane&7) ^ ((row>>1)&7) = (lane&7) ^ ((4i + (lane>>4)) & 7)
        l3, l7, l4, x, off = v[3], v[4], v[5], v[6], v[7]
        e("v_lshrrev_b32 %s, 3, %s" % (R("v", l3), R("v", lane)))
        e("v_and_b32 %s, 7, %s" % (R("v", l7), R("v", lane)))
        e("v_lshrrev_b32 %s, 4, %s" % (R("v", l4), R("v", lane)))
        for var in range(2):
            e("v_xor_b32 %s, %d, %s" % (R("v", x), 4 * var, R("v", l4)), "(row >> 1) & 7")
            e("v_xor_b32 %s, %s, %s" % (R("v", x), R("v", l7), R("v", x)))
            e("v_lshlrev_b32 %s, 4, %s" % (R("v", x), R("v", x)))
            e("v_mov_b32 %s, %d" % (R("v", off), c.w_row))
            e("v_mad_u32_u24 %s, %s, %s, %s" % (R("v", self.vA_dma[var]), R("v", l3), R("v", off), R("v", x)))
        # ---- B DMA lane parts (variant ib = piece & 1): rr = 8*ib + (lane>>3); channel = (2*ib + (lane>>5))*8 + ((lane>>3)&3);
        #      chunk = (lane&7) ^ (4*ib + ((lane>>4)&3))
        l5, ch = v[8], v[9]
        e("v_lshrrev_b32 %s, 5, %s" % (R("v", l5), R("v", lane)))
        for ib in range(2):
            e("v_lshl_add_u32 %s, %s, 3, %d" % (R("v", ch), R("v", l5), 16 * ib))
            e("v_and_b32 %s, 3, %s" % (R("v", x), R("v", l3)))
            e("v_add_u32 %s, %s, %s" % (R("v", ch), R("v", ch), R("v", x)))
            e("v_mov_b32 %s, %d" % (R("v", x), c.w_row))
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", ch), R("v", ch), R("v", x)))
            e("v_and_b32 %s, 3, %s" % (R("v", x), R("v", l4)))
            e("v_or_b32 %s, %d, %s" % (R("v", x), 4 * ib, R("v", x)))
            e("v_xor_b32 %s, %s, %s" % (R("v", x), R("v", l7), R("v", x)))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", self.vB_dma[ib]), R("v", x), R("v", ch)))
        # ---- fragment read bases.  A: row = r (+16 m as immediate), chunk = (kg + 4kk) ^ ((r>>1)&7); B: row = 64w + r
        sw, cc = v[3], v[4]
        e("v_bfe_u32 %s, %s, 1, 3" % (R("v", sw), R("v", r)))
        e("v_xor_b32 %s, %s, %s" % (R("v", cc), R("v", kg), R("v", sw)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", cc), R("v", cc)))
        e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", cc), R("v", r), R("v", cc)))
        e("v_add_u32 %s, %d, %s" % (R("v", self.vA_rd[0]), c.ABASE, R("v", cc)))
        e("v_xor_b32 %s, 64, %s" % (R("v", self.vA_rd[1]), R("v", self.vA_rd[0])))
        e("s_lshl_b32 %s, %s, 13" % (R("s", t0), R("s", self.s_w)), "64 rows x 128 B")
        e("v_add_u32 %s, %s, %s" % (R("v", self.vB0[0]), R("s", t0), R("v", cc)))
        e("v_xor_b32 %s, 64, %s" % (R("v", self.vB0[1]), R("v", self.vB0[0])))
        # ---- output lane offset: row r of a fragment, this wave's 64 columns, 8 channels per lane group
        e("v_mov_b32 %s, %d" % (R("v", off), c.N * 2))
        e("v_mul_lo_u32 %s, %s, %s" % (R("v", x), R("v", r), R("v", off)))
        e("s_lshl_b32 %s, %s, 7" % (R("s", t0), R("s", self.s_w)), "64 columns x 2 B")
        e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", x), R("v", kg), R("v", x)))
        e("v_add_u32 %s, %s, %s" % (R("v", self.v_out), R("s", t0), R("v", x)))

        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b32 %s, %s" % (R("s", self.s_alo), R("s", ka + 0)))
        e("s_mov_b32 %s, %s" % (R("s", self.s_ahi), R("s", ka + 1)))
        e("s_mov_b32 %s, %s" % (R("s", self.s_outlo), R("s", ka + 4)))
        e("s_mov_b32 %s, %s" % (R("s", self.s_outhi), R("s", ka + 5)))
        e("s_mov_b32 %s, %s" % (R("s", self.s_mtiles), R("s", kb + 2)))
        # units of this workgroup
        e("s_mul_i32 %s, %s, %s" % (R("s", self.s_u), R("s", self.s_wg), R("s", kb + 1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_uend), R("s", self.s_u), R("s", kb + 1)))
        e("s_min_u32 %s, %s, %s" % (R("s", self.s_uend), R("s", self.s_uend), R("s", kb + 0)))
        e("s_cmp_lt_u32 %s, %s" % (R("s", self.s_u), R("s", self.s_uend)))
        lab = self.newlabel("work")
        e("s_cbranch_scc1 %s" % lab)
        e("s_endpgm")
        self.label(lab)
        lg = c.NTN.bit_length() - 1
        e("s_lshr_b32 %s, %s, %d" % (R("s", self.s_mt), R("s", self.s_u), lg))
        e("s_and_b32 %s, %s, %d" % (R("s", self.s_nt), R("s", self.s_u), c.NTN - 1))
        # descriptors
        e("s_mov_b32 %s, %d" % (R("s", self.srdA + 2), c.ROWS * c.K * 2))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdA + 3))
        e("s_mov_b32 %s, %s" % (R("s", self.srdB), R("s", ka + 2)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdB + 1), R("s", ka + 3)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdB + 2), c.N * c.w_row))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdB + 3))
        e("s_mov_b32 %s, %s" % (R("s", self.srdX), R("s", ka + 6)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdX + 1), R("s", ka + 7)))
        e("s_mov_b32 %s, 0x7fffffff" % R("s", self.srdX + 2))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdX + 3))
        # previous unit: none -> a null output window (stores dropped), its accumulator set zeroed below
        e("s_mov_b32 %s, %s" % (R("s", self.srdOp), R("s", self.s_outlo)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdOp + 1), R("s", self.s_outhi)))
        e("s_mov_b32 %s, 0" % R("s", self.srdOp + 2))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdOp + 3))
        e("s_lshl_b32 %s, %s, 9" % (R("s", t0), R("s", self.s_w)))
        e("s_add_u32 %s, %s, %d" % (R("s", t0), R("s", t0), c.SBASE))
        e("v_lshl_add_u32 %s, %s, 6, %s" % (R("v", self.v_st), R("v", self.v_kg), R("s", t0)), "(the first unit's idle epilogue adds zeros here)")
        # first loads: the planes of this unit's pixel tile, weight stages 0 .. NB-1 of this unit
        self.set_srdA(self.s_mt)
        e("s_mov_b32 %s, 0" % R("s", self.s_bcur))
        for k in range(4):
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_nt), k))
            e("s_and_b32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_t0), c.NTN - 1))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_bsrc[k]), R("s", self.s_t0), 256 * c.w_row))
        for st in range(c.NB):
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_fillw), R("s", self.s_ldsBw), st * 32768))
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_stg), R("s", self.s_bsrc[st // c.NCH]), (st % c.NCH) * 128))
            for i in range(8):
                for ins in self.b_piece_insts(i, self.s_fillw, self.s_stg):
                    e(ins)
            if st == 0:
                for pl in range(c.NCH):
                    for k in range(len(a_slots(c))):
                        for ins in self.a_slot_insts(k, pl):
                            e(ins)
        self.unit_scalars()
        # statistics scratch = 0; the idle accumulator set = 0 (the first unit's "previous unit" epilogue stores nothing and adds 0)
        e("v_mov_b32 %s, 0" % R("v", v[3]))
        e("v_lshlrev_b32 %s, 2, v0" % R("v", v[4]))
        e("v_add_u32 %s, %d, %s" % (R("v", v[4]), c.SBASE, R("v", v[4])))
        for k in range(c.NTN * 256 * 8 // 1024):
            e("ds_write_b32 %s, %s offset:%d" % (R("v", v[4]), R("v", v[3]), k * 1024))
        for i in range(self.nacc):
            e("v_accvgpr_write_b32 a%d, 0" % (self.nacc + i))
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        # fragments of (stage 0, kk 0): ring offset 0
        e("v_mov_b32 %s, %s" % (R("v", self.vBnext0), R("v", self.vB0[0])))
        for ins in self.frag_reads(0, 0, 0):
            e(ins)

    def frag_reads(self, fset, plane, kk):
        """fragments of (plane, kk): A from the plane, B from the current stage (kk = 1) or the next stage (kk = 0)"""
        c = self.c
        fa, fb = self.F[fset]
        out = []
        vb = self.vBcur1 if kk == 1 else self.vBnext0
        order = []
        for n in range(4):
            order.append(("b", n))
            order.append(("a", n))
        for m in range(4, c.MT):
            order.append(("a", m))
        for kind, i in order:
            if kind == "a":
                out.append("ds_read_b128 %s, %s offset:%d" % (R("v", fa + 4 * i, 4), R("v", self.vA_rd[kk]), plane * c.PLANE + i * 2048))
            else:
                out.append("ds_read_b128 %s, %s offset:%d" % (R("v", fb + 4 * i, 4), R("v", vb), i * 2048))
        return out

    def mfmas(self, fset, ap, zero_c):
        c = self.c
        fa, fb = self.F[fset]
        out = []
        for n in range(4):
            for m in range(c.MT):
                acc = ap * self.nacc + (m * 4 + n) * 4
                src = "0" if zero_c else R("a", acc, 4)
                out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (R("a", acc, 4), R("v", fb + 4 * n, 4), R("v", fa + 4 * m, 4), src))
        return out

    # ---- epilogue of one accumulator set, as a list of instruction groups --------------------------------------------------
    def epi_item(self, ap, p, m, k):
        """fragment m, tile pair p of accumulator set ap: read, convert, store (+ statistics); k = running item index"""
        c = self.c
        tv, xr, s1, s2 = self.tv, self.xr, self.s1, self.s2
        d = self.dsets[k % 4]
        g = []
        for i in range(4):
            if c.probe & 32:
                break
            g.append("v_accvgpr_read_b32 %s, a%d" % (R("v", tv[i]), ap * self.nacc + (m * 4 + 2 * p) * 4 + i))
            g.append("v_accvgpr_read_b32 %s, a%d" % (R("v", tv[4 + i]), ap * self.nacc + (m * 4 + 2 * p + 1) * 4 + i))
        if c.probe & 16:
            return g
        for i in range(4):
            g.append("v_cvt_pk_bf16_f32 %s, %s, %s" % (R("v", d + i), R("v", tv[2 * i]), R("v", tv[2 * i + 1])))
        g.append("s_mov_b32 %s, %d" % (R("s", self.s_t1), m * 16 * c.N * 2))
        if not (c.probe & 2):
            g.append(("vm", "store", "buffer_store_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("v", d, 4), R("v", self.v_out), R("s", self.srdOp, 4), R("s", self.s_t1), p * 64)))
        if c.stats:
            for i in range(4):
                g.append("v_lshlrev_b32 %s, 16, %s" % (R("v", xr[2 * i]), R("v", d + i)))
                g.append("v_and_b32 %s, 0xffff0000, %s" % (R("v", xr[2 * i + 1]), R("v", d + i)))
            for i in range(8):
                if m == 0:  # first fragment of the pair: start the sums
                    g.append("v_mov_b32 %s, %s" % (R("v", s1[i]), R("v", xr[i])))
                    g.append("v_mul_f32 %s, %s, %s" % (R("v", s2[i]), R("v", xr[i]), R("v", xr[i])))
                else:
                    g.append("v_add_f32 %s, %s, %s" % (R("v", s1[i]), R("v", s1[i]), R("v", xr[i])))
                    g.append("v_fma_f32 %s, %s, %s, %s" % (R("v", s2[i]), R("v", xr[i]), R("v", xr[i]), R("v", s2[i])))
        return g

    def epi_stat_finish(self, p):
        """row sums of the pair's 16 statistics registers, then lanes 15 add them into the workgroup's LDS rows (one contiguous
        group: EXEC is narrowed inside it)"""
        c = self.c
        g = []
        for sh in (1, 2, 4, 8):
            for arr in (self.s1, self.s2):
                for i in range(8):
                    rr = R("v", arr[i])
                    g.append("v_add_f32_dpp %s, %s, %s row_shr:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (rr, rr, rr, sh))
        tail = ["s_mov_b32 exec_lo, 0x80008000", "s_mov_b32 exec_hi, 0x80008000"]
        for i in range(8):
            tail.append("ds_add_f32 %s, %s offset:%d" % (R("v", self.v_st), R("v", self.s1[i]), p * 32 * 8 + i * 8))
            tail.append("ds_add_f32 %s, %s offset:%d" % (R("v", self.v_st), R("v", self.s2[i]), p * 32 * 8 + i * 8 + 4))
        tail.append("s_mov_b64 exec, -1")
        return g, tail

    def epi_groups(self, ap):
        """the whole epilogue of accumulator set ap as (list of per-instruction entries, positions of contiguous blocks)"""
        c = self.c
        items = []
        k = 0
        for p in range(2):
            for m in range(c.MT):
                items.append(("item", self.epi_item(ap, p, m, k)))
                k += 1
            if c.stats:
                g, tail = self.epi_stat_finish(p)
                items.append(("item", g))
                items.append(("block", tail))
        return items

    def interleave(self, mf, groups, skip=None):
        """groups: list of instruction lists kept together; spread evenly between the MFMAs.  skip: groups (by identity) left out
        WITHOUT moving the others (the two copies of a substep must issue their common instructions in the same order)"""
        n, k = len(mf), len(groups)
        slots = {}
        for j, grp in enumerate(groups):
            pos = (j * n) // k if k else 0
            slots.setdefault(pos, []).append(grp)
        skip_ids = {id(g) for g in (skip or [])}
        for i, m in enumerate(mf):
            self.e(m)
            for grp in slots.get(i, []):
                if id(grp) not in skip_ids:
                    self.emit_group(grp)

    @staticmethod
    def chop(entries, size):
        """split a flat instruction list into groups of `size`"""
        return [entries[i:i + size] for i in range(0, len(entries), size)]

    def wait_vm(self, required):
        slot = len(self.wait_slots)
        self.wait_slots[slot] = required
        self.tr.wait(slot, required)
        self.e("s_waitcnt vmcnt(@VM%d@)" % slot)

    # -----------------------------------------------------------------------------------------------------------------
    def unit_body(self, ub):
        """unit `ub` of the loop body accumulating into set ub & 1, with the epilogue of the other set (the previous unit) between its MFMAs"""
        ap = ub & 1
        c, e = self.c, self.e
        nsub = self.UB * c.NCH
        # distribute the previous unit's epilogue over the substeps: flat stream of entries, blocks stay whole
        epi = self.epi_groups(ap ^ 1) if not (c.probe & 1) else []
        flat = []   # list of groups (each a list of instructions kept together)
        for kind, g in epi:
            if kind == "block":
                flat.append(list(g))
            else:
                flat.extend(self.chop(g, 3))
        per = [flat[(len(flat) * s) // nsub:(len(flat) * (s + 1)) // nsub] for s in range(nsub)]
        for ch in range(c.NCH):
            # ---- substep kk = 0
            self.comment("acc set %d stage %d substep 0" % (ap, ch))
            e("v_add_u32 %s, %s, %s" % (R("v", self.vBcur1), R("s", self.s_bcur), R("v", self.vB0[1])))
            e("s_waitcnt lgkmcnt(0)")
            groups = [[r] for r in self.frag_reads(1, ch, 1)]
            self.interleave(self.mfmas(0, ap, zero_c=(ch == 0)), self.merge(per[2 * ch], groups))
            # ---- the stage barrier: the next stage's weights have landed for every wave (and, at the last stage, plane 0)
            self.comment("acc set %d stage %d substep 1" % (ap, ch))
            g = (ub * c.NCH + ch)
            need = [("B", (g + 1) % (self.UB * c.NCH))]
            nplane = (ch + 1) % c.NCH
            need.append(("A", nplane, (g + 1) % (self.UB * c.NCH) // c.NCH))
            self.wait_vm(need)
            e("s_waitcnt lgkmcnt(0)")
            if not (c.probe & 8):
                e("s_barrier")
            # ring: next stage's offset; this stage's slot is refilled below
            e("s_add_u32 %s, %s, 32768" % (R("s", self.s_bnext), R("s", self.s_bcur)))
            e("s_cmp_eq_u32 %s, %d" % (R("s", self.s_bnext), c.NB * 32768))
            e("s_cselect_b32 %s, 0, %s" % (R("s", self.s_bnext), R("s", self.s_bnext)))
            e("v_add_u32 %s, %s, %s" % (R("v", self.vBnext0), R("s", self.s_bnext), R("v", self.vB0[0])))
            e("s_add_u32 %s, %s, %s" % (R("s", self.s_fillw), R("s", self.s_ldsBw), R("s", self.s_bcur)))
            # weight stage + NB: chunk (ch + NB) % NCH of this unit or the next
            cf = (ch + c.NB) % c.NCH
            src = self.s_bsrc[(ch + c.NB) // c.NCH]
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_stg), R("s", src), cf * 128))
            groups = [[r] for r in self.frag_reads(0, nplane, 0)]
            dma = []
            gt = (g + c.NB) % (self.UB * c.NCH)
            for i in range(8):
                ins = self.b_piece_insts(i, self.s_fillw, self.s_stg)
                dma.append(ins[:2] + [("vm", ("B", gt), ins[2])])
            if c.probe & 4:
                dma = [[("vm", ("B", gt), "s_nop 0")] for i in range(8)]
            mf = self.mfmas(1, ap, zero_c=False)
            # A refill of plane ch with the NEXT pixel tile — only during the last column tile of this pixel tile: two copies of
            # the substep, selected by a scalar branch (same MFMAs / reads / weight pieces in both)
            lab_no, lab_end = self.newlabel("norefill"), self.newlabel("joined")
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_last))
            e("s_cbranch_scc1 %s" % lab_no)
            adma = []
            for k in range(len(a_slots(c))):
                ins = self.a_slot_insts(k, ch)
                adma.append(ins[:2] + [("vm", ("A", ch, (ub + 1) % self.UB), ins[2], True)])
            saved = list(self.tr.events)
            allg = self.merge(self.merge(per[2 * ch + 1], groups), self.merge(dma, adma))
            self.interleave(mf, allg)
            e("s_branch %s" % lab_end)
            self.label(lab_no)
            # (the tracker follows the refill path: its extra instructions are flagged conditional; the other path is the same
            # stream without them)
            ev_refill = self.tr.events
            self.tr.events = list(saved)
            self.interleave(mf, allg, skip=adma)
            self.tr.events = ev_refill
            self.label(lab_end)
            e("s_mov_b32 %s, %s" % (R("s", self.s_bcur), R("s", self.s_bnext)))

    def emit_group(self, entries):  # (redefined: conditional flag support)
        for ins in entries:
            if isinstance(ins, tuple):
                tag, text = ins[1], ins[2]
                self.tr.op(tag, conditional=len(ins) > 3 and ins[3])
                self.e(text)
            else:
                self.e(ins)

    @staticmethod
    def merge(a, b):
        if not b:
            return list(a)
        if not a:
            return list(b)
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

    def unit_switch(self, ap, lab_exit):
        """the unit that just ran becomes the previous one; advance to the next unit or leave the loop"""
        c, e = self.c, self.e
        tb = c.ROWS * c.N * 2
        t0, t1 = self.s_t0, self.s_t1
        # previous unit's output window: out + mt*ROWS*N*2 + nt*512, records = ROWS*N*2 - nt*512
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_mt), tb))
        e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_mt), tb))
        e("s_lshl_b32 %s, %s, 9" % (R("s", self.s_stg), R("s", self.s_nt)))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", self.s_stg)))
        e("s_addc_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdOp), R("s", self.s_outlo), R("s", t0)))
        e("s_addc_u32 %s, %s, %s" % (R("s", self.srdOp + 1), R("s", self.s_outhi), R("s", t1)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdOp + 1), R("s", self.srdOp + 1)))
        e("s_sub_u32 %s, %d, %s" % (R("s", self.srdOp + 2), tb, R("s", self.s_stg)))
        # statistics rows of the previous unit's column tile: this lane's slot = SBASE + (pnt*256 + 64w + kg*8)*8
        e("s_lshl_b32 %s, %s, 11" % (R("s", t0), R("s", self.s_nt)))
        e("s_lshl_b32 %s, %s, 9" % (R("s", t1), R("s", self.s_w)))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t1)))
        e("s_add_u32 %s, %s, %d" % (R("s", t0), R("s", t0), c.SBASE))
        e("v_lshl_add_u32 %s, %s, 6, %s" % (R("v", self.v_st), R("v", self.v_kg), R("s", t0)))
        # next unit
        e("s_add_u32 %s, %s, 1" % (R("s", self.s_u), R("s", self.s_u)))
        e("s_cmp_lt_u32 %s, %s" % (R("s", self.s_u), R("s", self.s_uend)))
        e("s_cbranch_scc0 %s" % lab_exit)
        e("s_mov_b32 %s, %s" % (R("s", self.s_nt), R("s", self.s_ntn)))
        e("s_cmp_eq_u32 %s, 0" % R("s", self.s_nt))
        e("s_addc_u32 %s, %s, 0" % (R("s", self.s_mt), R("s", self.s_mt)))
        self.unit_scalars()

    def loop(self):
        c, e = self.c, self.e
        top = self.newlabel("units")
        exits = [self.newlabel("exit0"), self.newlabel("exit1")]
        done = self.newlabel("done")
        self.label(top)
        for ub in range(self.UB):
            self.unit_body(ub)
            self.unit_switch(ub & 1, exits[ub & 1])
        e("s_branch %s" % top)
        # ---- the last unit's epilogue, not interleaved
        for ap in range(2):
            self.label(exits[ap])
            e("s_nop 15")
            e("s_nop 15")
            saved = self.tr.events
            self.tr.events = []
            for kind, g in self.epi_groups(ap):
                self.emit_group(g)
            self.tr.events = saved
            if ap == 0:
                e("s_branch %s" % done)
        self.label(done)
        # ---- statistics row of this workgroup: row[c] = sum, row[N + c] = sum of squares, c = 0 .. N-1
        if c.stats:
            e("s_waitcnt lgkmcnt(0)")
            e("s_barrier")
            v = self.v_t
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_wg), 2 * c.N * 4))
            e("s_add_u32 %s, %s, %d" % (R("s", self.s_t1), R("s", self.s_t0), c.N * 4))
            e("v_lshlrev_b32 %s, 3, v0" % R("v", v[0]))
            e("v_add_u32 %s, %d, %s" % (R("v", v[0]), c.SBASE, R("v", v[0])))
            e("v_lshlrev_b32 %s, 2, v0" % R("v", v[1]))
            for k in range(c.NTN):
                e("ds_read_b64 %s, %s offset:%d" % (R("v", self.tv[0], 2), R("v", v[0]), k * 2048))
                e("s_waitcnt lgkmcnt(0)")
                e("buffer_store_dword %s, %s, %s, %s offen offset:%d" % (R("v", self.tv[0]), R("v", v[1]), R("s", self.srdX, 4), R("s", self.s_t0), k * 1024))
                e("buffer_store_dword %s, %s, %s, %s offen offset:%d" % (R("v", self.tv[1]), R("v", v[1]), R("s", self.srdX, 4), R("s", self.s_t1), k * 1024))
                e("s_nop 1")
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")

    # -----------------------------------------------------------------------------------------------------------------
    def finish(self):
        c = self.c
        name = c.name
        total_v = self.accum_offset + self.nagpr
        assert total_v <= 512
        hdr = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6", "\t.text", "\t.protected\t%s" % name,
               "\t.globl\t%s" % name, "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
        tail = ["\t.section\t.rodata,\"a\",@progbits", "\t.p2align\t6, 0x0", "\t.amdhsa_kernel %s" % name]
        kd = dict(group_segment_fixed_size=c.LDS, private_segment_fixed_size=0, kernarg_size=self.KA["size"],
                  user_sgpr_count=2, user_sgpr_dispatch_ptr=0, user_sgpr_queue_ptr=0, user_sgpr_kernarg_segment_ptr=1,
                  user_sgpr_dispatch_id=0, user_sgpr_kernarg_preload_length=0, user_sgpr_kernarg_preload_offset=0,
                  user_sgpr_private_segment_size=0, uses_dynamic_stack=0, enable_private_segment=0,
                  system_sgpr_workgroup_id_x=1, system_sgpr_workgroup_id_y=0, system_sgpr_workgroup_id_z=0,
                  system_sgpr_workgroup_info=0, system_vgpr_workitem_id=0, next_free_vgpr=total_v,
                  next_free_sgpr=self.S.n, accum_offset=self.accum_offset, reserve_vcc=1, float_round_mode_32=0,
                  float_round_mode_16_64=0, float_denorm_mode_32=3, float_denorm_mode_16_64=3, dx10_clamp=1, ieee_mode=1,
                  fp16_overflow=0, tg_split=0)
        for k, v in kd.items():
            tail.append("\t\t.amdhsa_%s %d" % (k, v))
        tail += ["\t.end_amdhsa_kernel", "\t.text", "\t.amdgpu_metadata", "---", "amdhsa.kernels:", "  - .agpr_count:     %d" % self.nagpr,
                 "    .args:"]
        off = 0
        for i in range(10):
            tail.append("      - .address_space:  global\n        .offset:         %d\n        .size:           8\n        .value_kind:     global_buffer" % off)
            off += 8
        tail.append("      - .offset:         %d\n        .size:           %d\n        .value_kind:     by_value" % (off, self.KA["size"] - off))
        tail += ["    .group_segment_fixed_size: %d" % c.LDS, "    .kernarg_segment_align: 8", "    .kernarg_segment_size: %d" % self.KA["size"],
                 "    .max_flat_workgroup_size: 256", "    .name:           %s" % name, "    .private_segment_fixed_size: 0",
                 "    .sgpr_count:     %d" % (self.S.n + 6), "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % name,
                 "    .uniform_work_group_size: 1", "    .uses_dynamic_stack: false", "    .vgpr_count:     %d" % total_v,
                 "    .vgpr_spill_count: 0", "    .wavefront_size: 64", "amdhsa.target:   amdgcn-amd-amdhsa--gfx950",
                 "amdhsa.version:\n  - 1\n  - 2", "...", "\t.end_amdgpu_metadata"]
        body = self.out + ["\t.p2align 8", ".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name)]
        return "\n".join(hdr + body + tail) + "\n"


VARIANTS = {
    "pw_k256_n1024_s1": PwCfg("pw_k256_n1024_s1", K=256, N=1024, stats=1),
    "pw_k256_n1024_s0": PwCfg("pw_k256_n1024_s0", K=256, N=1024, stats=0),
    # (K = 128 -> 512, layer 2's conv3, is generated correctly too — tests/test_dconv_emu.py — and measured no faster than the
    # 2-workgroups-per-CU kernel, 80 vs 81 us: store-bound, and one wave per SIMD keeps too few stores in flight; not shipped)
}


def generate(base, **over):
    c = VARIANTS[base]
    if over:
        c = PwCfg(**{**c.__dict__, **over})
    g = Gen(c)
    text = g.gen()
    return c, g, text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="build")
    ap.add_argument("--set", action="append", default=[])
    ap.add_argument("--suffix", default="")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    over = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.set}
    for name in (a.names or VARIANTS):
        if a.suffix:
            over["name"] = name + a.suffix
        c, g, text = generate(name, **over)
        if a.suffix:
            with open(os.path.join(a.out, c.name + ".tbl"), "wb") as f:
                f.write(struct.pack("<128I", *[w for row in tables(c) for w in row]))
        with open(os.path.join(a.out, c.name + ".s"), "w") as f:
            f.write(text)
        print("%s: %d lines, %d VGPR + %d AGPR, %d SGPR, LDS %d" % (c.name, text.count("\n"), g.accum_offset, g.nagpr, g.S.n, c.LDS))


if __name__ == "__main__":
    main()
