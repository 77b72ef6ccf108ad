#!/usr/bin/env python3
"""pk_gen.py — generator of the hand-scheduled gfx950 (MI355X) pointwise convolution kernels with a LONG reduction (K = 1024 / 2048
input channels, 256 output columns per workgroup): conv1 of the bottlenecks of ResNet-50's layers 3 and 4 (forward, with the BN
statistics rows) and the data gradient of their conv3 (with the BN-backward sums of bn2) — the IgemmArgs contract of launch_igemm()
(conv forward / data gradient under `model(data)` / `loss.backward()`, /root/reference/sota_imagenet/callbacks.py:316-317).

Why its own structure: these launches read 4x more than they write and carry only 26 GFLOP over 129 MB (layer 3): 14 us of MFMA
against 23 us of HBM.  What bounds them is bytes in flight per CU, so the kernel is the direct-convolution kernel of dconv_gen.py
(one wave per SIMD, accumulators in AGPRs, fragment reads and LDS-DMA issue between the MFMAs, one barrier per stage, its epilogues
reused as they are) with the 9-tap loop replaced by a walk over 64-channel chunks and
  * THREE activation buffers: the tile of chunk c + 2 is requested in the first substep of chunk c and has until the barrier of
    chunk c + 1 to land (the implicit-GEMM kernels keep one stage in flight; pw_gen.py keeps K resident, which 1024 channels are not),
  * the weight slab of chunk c + 2 requested behind the barrier of chunk c (2- or 3-stage ring, L2-resident),
  * counted waits: one activation tile stays in flight across every barrier.
A tile is W consecutive pixels (196 = one 14 x 14 image; 98 = two 7 x 7 images, or half an image where the BN-backward register sets
need the shorter tile) x 256 columns; waves 1 (M) x 4 (N).
"""
import argparse
import os
import sys
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dconv_gen  # noqa: E402
from dconv_gen import Alloc, R  # noqa: E402


@dataclass
class PkCfg(dconv_gen.Cfg):
    NA_: int = 3      # activation buffers

    @property
    def NA(self):
        return self.NA_

    @property
    def LROWS(self):
        return 1

    @property
    def APIECES(self):    # 1 KiB LDS-DMA pieces of an activation tile per wave (whole blocks of 8 pixels, 4 waves)
        return -(-self.W // 32)

    @property
    def ABUF(self):
        return self.APIECES * 4 * 1024

    @property
    def ASTRIDE(self):
        return self.ABUF

    @property
    def w_row(self):      # bytes of one weight row [1][Cin]
        return self.Cin * 2


def mk(name, W, Cin, NCOLS, stats, NB=2, NA=3, NT=4):
    P = (W + 15) // 16 * 16
    return PkCfg(name, H=1, W=W, P=P, IPT=1, Cin=Cin, NCOLS=NCOLS, stats=stats, WM=1, WN=4, NT=NT, NB=NB, NA_=NA)


class Gen(dconv_gen.Gen):
    KA = dict(in_=0, wt=8, out=16, stat=24, bn_y=32, bn_bits=40, bn_mean=48, bn_invstd=56, rsvd=64, nchunks=72, size=128)

    def gen(self):
        c = self.c
        S, V = self.S, self.V
        assert c.WM == 1 and c.WN == 4 and c.NT in (2, 4) and c.BN in (128, 256) and c.H == 1 and c.IPT == 1 and not c.ROWS_T
        self.NPA = c.APIECES
        self.s_tile, self.s_nt = 2, 3
        self.srdA = S.get(4, 4)
        self.srdB = S.get(4, 4)
        self.srdO = S.get(4, 4)
        self.srdX = S.get(4, 4)
        if c.stats >= 2:
            self.srdM = S.get(4, 4)
            self.srdY = S.get(4, 4)
            self.srdMu = S.get(4, 4)
            self.srdIs = S.get(4, 4)
        (self.s_cnt, self.s_cA, self.s_cB, self.s_clast, self.s_t0, self.s_t1, self.s_w, self.s_wm, self.s_wn, self.s_nch, self.s_ldsBw, self.s_ldsAw,
         self.s_stg, self.s_srcAw, self.s_srcBw) = [S.get() for _ in range(15)]
        self.s_ka = S.get(16, 4)
        self.s_kb = S.get(4, 4)
        self.v_tid = 0
        self.vA_rd = [[V.get() for kk in range(2)] for b in range(c.NA)]
        self.vB_rd = [[V.get() for kk in range(2)] for st in range(c.NB)]
        self.vA_dma = V.get()
        self.vB_dma = [V.get() for _ in range(2)]
        self.v_out = V.get()
        self.v_kg = V.get()
        self.v_tmp = [V.get(), V.get()]
        self.F = []
        for s in range(2):
            fa = V.get(4 * c.MFR, 4)
            fb = V.get(4 * c.NT, 4)
            self.F.append((fa, fb))
        self.v_t = [self.F[1][0] + i for i in range(10)]
        self.v_t[2] = self.v_kg
        self.late_pair1 = False
        if c.stats >= 2:
            # BN-backward inputs of the two tile pairs.  Where two full sets do not fit (13 fragments), the second set is loaded at the
            # start of the epilogue INTO the fragment registers (free by then; the epilogue's temporaries use the first 51 of them)
            self.late_pair1 = c.MFR > 8 and c.NT >= 4
            if c.NT < 4:  # one tile pair per wave: one register set
                y0, m0 = [V.get(4, 4) for m in range(c.MFR)], V.get(16, 4)
                self.ysets, self.msets = [y0, y0], [m0, m0]
            elif self.late_pair1:
                f0, f1 = self.F[0][0], self.F[1][0]
                free = list(range(f0 + 52, f0 + 4 * c.MFR + 16)) + list(range(f1, f1 + 4 * c.MFR + 16))
                free4 = [r for r in free if r % 4 == 0 and all(r + i in free for i in range(4))]
                y1 = free4[:c.MFR]
                used = {r + i for r in y1 for i in range(4)}
                rest4 = [r for r in free4 if r not in used]
                m1 = rest4[:4]
                assert len(y1) == c.MFR and len(m1) == 4 and m1 == list(range(m1[0], m1[0] + 16, 4)), "no room for the late register set"
                self.ysets = [[V.get(4, 4) for m in range(c.MFR)], y1]
                self.msets = [V.get(16, 4), m1[0]]
            else:
                self.ysets = [[V.get(4, 4) for m in range(c.MFR)] for _ in range(2)]
                self.msets = [V.get(16, 4) for _ in range(2)]
            self.v_chan = V.get()
            self.alloc_tile_masks()
        self.nvgpr = V.n
        self.accum_offset = (self.nvgpr + 7) // 8 * 8
        self.nagpr = c.MFR * c.NT * 4
        assert self.nagpr <= 256 and self.accum_offset + self.nagpr <= 512
        self.tmp_i = 0
        self.prologue()
        self.mainloop()
        self.epilogue()
        return self.finish()

    # -----------------------------------------------------------------------------------------------------------------
    def a_piece(self, k, buf, s_chunk):
        """activation piece k of this wave (8-pixel block 4k + w of the tile) of the chunk at byte offset s_chunk into buffer buf"""
        c = self.c
        vt = self.v_tmp[self.tmp_i & 1]
        self.tmp_i += 1
        return ["s_add_u32 m0, %s, %d" % (R("s", self.s_ldsAw), c.ABASE + buf * c.ASTRIDE + k * 4096),
                "s_add_u32 %s, %s, %d" % (R("s", self.s_t1), R("s", s_chunk), k * 32 * c.Cin * 2),
                "v_add_u32 %s, %s, %s" % (R("v", vt), R("s", self.s_t1), R("v", self.vA_dma)),
                "buffer_load_dwordx4 %s, %s, 0 offen lds" % (R("v", vt), R("s", self.srdA, 4))]

    @staticmethod
    def b_const(c, i):
        """source constant of weight piece i of a wave (rows permuted so that a lane's tile pair is 8 consecutive channels: the
        b_piece_const of dconv_gen.py without its wave part)"""
        return ((i >> 2) * 32 + 4 * ((i >> 1) & 1)) * c.w_row

    def b_piece(self, i, bp, s_chunk):
        c = self.c
        return ["s_add_u32 m0, %s, %d" % (R("s", self.s_ldsBw), c.BBASE + bp * c.BSTAGE + i * 1024),
                "s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", s_chunk), self.b_const(c, i)),
                "buffer_load_dwordx4 %s, %s, %s offen lds" % (R("v", self.vB_dma[i & 1]), R("s", self.srdB, 4), R("s", self.s_t0))]

    def advance(self, s):
        """s = min(s + 128, last chunk offset): loads past the last chunk re-read it (never used)"""
        return ["s_add_u32 %s, %s, 128" % (R("s", s), R("s", s)), "s_min_u32 %s, %s, %s" % (R("s", s), R("s", s), R("s", self.s_clast))]

    def frag_reads(self, fset, kk, abuf, bst):
        c = self.c
        fa, fb = self.F[fset]
        out = []
        order = []
        for n in range(c.NT):
            order.append(("b", n))
            if n < c.MFR:
                order.append(("a", n))
        for m in range(c.NT, c.MFR):
            order.append(("a", m))
        for kind, i in order:
            if kind == "a":
                out.append("ds_read_b128 %s, %s offset:%d" % (R("v", fa + 4 * i, 4), R("v", self.vA_rd[abuf][kk]), i * 2048))
            else:
                out.append("ds_read_b128 %s, %s offset:%d" % (R("v", fb + 4 * i, 4), R("v", self.vB_rd[bst][kk]), i * 2048))
        return out

    # -----------------------------------------------------------------------------------------------------------------
    def prologue(self):
        c, e = self.c, self.e
        ka, kb = self.s_ka, self.s_kb
        v = self.v_t
        t0, t1 = self.s_t0, self.s_t1
        self.comment("---- prologue: kernel arguments, lane constants, descriptors, first loads")
        e("s_load_dwordx16 %s, s[0:1], 0x0" % R("s", ka, 16))
        e("s_load_dword %s, s[0:1], 0x48" % R("s", kb))            # nchunks
        lane, r, kg = v[0], v[1], v[2]
        e("v_lshrrev_b32 %s, 6, v0" % R("v", v[3]))
        e("v_and_b32 %s, 63, v0" % R("v", lane))
        e("v_readfirstlane_b32 %s, %s" % (R("s", self.s_w), R("v", v[3])))
        e("v_and_b32 %s, 15, v0" % R("v", r))
        e("v_bfe_u32 %s, v0, 4, 2" % R("v", kg))
        e("s_nop 3")
        e("s_mov_b32 %s, 0" % R("s", self.s_wm))
        e("s_mov_b32 %s, %s" % (R("s", self.s_wn), R("s", self.s_w)))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_ldsBw), R("s", self.s_w), c.BSTAGE // 4), "this wave's quarter of a weight stage")
        e("s_lshl_b32 %s, %s, 10" % (R("s", self.s_ldsAw), R("s", self.s_w)), "this wave's 8-pixel block of every 32 pixels")
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_srcAw), R("s", self.s_w), 8 * c.Cin * 2))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_srcBw), R("s", self.s_w), c.NT * 16 * c.w_row))
        # ---- DMA lane parts.  A: row-in-block = lane >> 3; logical chunk = (lane & 7) ^ ((row >> 1) & 7), row = (4k + w)*8 + (lane >> 3)
        l3, l7, j, x, off = v[3], v[4], v[5], v[6], v[7]
        e("v_lshrrev_b32 %s, 3, %s" % (R("v", l3), R("v", lane)))
        e("v_and_b32 %s, 7, %s" % (R("v", l7), R("v", lane)))
        e("v_lshrrev_b32 %s, 4, %s" % (R("v", j), R("v", lane)), "(row >> 1) & 3 of the block-local row")
        e("s_and_b32 %s, %s, 1" % (R("s", t0), R("s", self.s_w)))
        e("s_lshl_b32 %s, %s, 2" % (R("s", t0), R("s", t0)), "+ 4 for the odd blocks (block = 4k + w)")
        e("v_or_b32 %s, %s, %s" % (R("v", j), R("s", t0), R("v", j)))
        e("v_xor_b32 %s, %s, %s" % (R("v", j), R("v", l7), R("v", j)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", j), R("v", j)))
        e("v_mov_b32 %s, %d" % (R("v", off), c.Cin * 2))
        e("v_mad_u32_u24 %s, %s, %s, %s" % (R("v", self.vA_dma), R("v", l3), R("v", off), R("v", j)))
        # B (variant ib = piece & 1), as dconv_gen.py
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
        e("s_waitcnt lgkmcnt(0)")
        self.comment("descriptors: A = this tile's pixels, B = this column tile's weight rows, O = this tile's output pixels")
        tile_in = c.W * c.Cin * 2
        tile_out = c.W * c.NCOLS * 2
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_tile), tile_in))
        e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_tile), tile_in))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdA), R("s", ka + 0), R("s", t0)))
        e("s_addc_u32 %s, %s, %s" % (R("s", self.srdA + 1), R("s", ka + 1), R("s", t1)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdA + 1), R("s", self.srdA + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdA + 2), tile_in), "pixels past the tile read zeros")
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdA + 3))
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_nt), c.BN * c.w_row))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdB), R("s", ka + 2), R("s", t0)))
        e("s_addc_u32 %s, %s, 0" % (R("s", self.srdB + 1), R("s", ka + 3)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdB + 1), R("s", self.srdB + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdB + 2), c.BN * c.w_row))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdB + 3))
        e("s_mov_b32 %s, %s" % (R("s", self.s_nch), R("s", kb)))
        e("s_sub_u32 %s, %s, 1" % (R("s", self.s_clast), R("s", self.s_nch)))
        e("s_lshl_b32 %s, %s, 7" % (R("s", self.s_clast), R("s", self.s_clast)), "byte offset of the last chunk in a pixel / weight row")

        def first_loads():
            self.comment("first loads: weight stage 0, activation tiles 0 and 1, the other weight stages")
            # s_cA / s_cB = this wave's source base + the chunk offset of the NEXT chunk to request
            e("s_mov_b32 %s, %s" % (R("s", self.s_cB), R("s", self.s_srcBw)))
            e("s_mov_b32 %s, %s" % (R("s", self.s_cA), R("s", self.s_srcAw)))
            for i in range(c.NPB):
                for ins in self.b_piece(i, 0, self.s_cB):
                    e(ins)
            self.adv_b()
            for b in range(2):
                for k in range(self.NPA):
                    for ins in self.a_piece(k, b, self.s_cA):
                        e(ins)
                self.adv_a()
            for st in range(1, c.NB):
                for i in range(c.NPB):
                    for ins in self.b_piece(i, st, self.s_cB):
                        e(ins)
                self.adv_b()

        def descriptors_out():
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
            e("s_sub_u32 %s, %s, %s" % (R("s", self.srdO + 2), R("s", self.srdO + 2), R("s", self.s_stg)))
            e("s_mov_b32 %s, %s" % (R("s", self.srdX), R("s", ka + 6)), "statistics rows")
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdX + 1), R("s", ka + 7)))
            e("s_mov_b32 %s, 0x7fffffff" % R("s", self.srdX + 2))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdX + 3))
            if c.stats >= 2:
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
            x, off = v[6], v[7]
            e("v_mov_b32 %s, %d" % (R("v", off), c.NCOLS * 2))
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", x), R("v", r), R("v", off)))
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 2))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", x), R("v", kg), R("v", x)))
            e("v_add_u32 %s, %s, %s" % (R("v", self.v_out), R("s", t0), R("v", x)))

        if c.stats >= 2:
            descriptors_out()
            lane_out()
            self.tile_mask_loads()
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 4))
            e("v_lshl_add_u32 %s, %s, 5, %s" % (R("v", self.v_chan), R("v", self.v_kg), R("s", t0)), "this lane's 8 floats of mean / invstd")
            self.epi_issue_loads(0)
            if not self.late_pair1 and c.NT >= 4:
                self.epi_issue_loads(1)
            first_loads()
        else:
            first_loads()
            descriptors_out()
            lane_out()
        # ---- read bases.  A: row r of a fragment, chunk (kg + 4kk) ^ ((r >> 1) & 7) (16 rows per fragment: the swizzle repeats);  B as dconv_gen.py
        sw, cc = v[3], v[4]
        e("v_bfe_u32 %s, %s, 1, 3" % (R("v", sw), R("v", r)))
        e("v_xor_b32 %s, %s, %s" % (R("v", cc), R("v", kg), R("v", sw)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", cc), R("v", cc)))
        e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", cc), R("v", r), R("v", cc)))
        for b in range(c.NA):
            e("v_add_u32 %s, %d, %s" % (R("v", self.vA_rd[b][0]), c.ABASE + b * c.ASTRIDE, R("v", cc)))
            e("v_xor_b32 %s, 64, %s" % (R("v", self.vA_rd[b][1]), R("v", self.vA_rd[b][0])))
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 128))
        e("s_add_u32 %s, %s, %d" % (R("s", t0), R("s", t0), c.BBASE))
        e("v_add_u32 %s, %s, %s" % (R("v", self.vB_rd[0][0]), R("s", t0), R("v", cc)))
        e("v_xor_b32 %s, 64, %s" % (R("v", self.vB_rd[0][1]), R("v", self.vB_rd[0][0])))
        for st in range(1, c.NB):
            for kk in range(2):
                e("v_add_u32 %s, %d, %s" % (R("v", self.vB_rd[st][kk]), st * c.BSTAGE, R("v", self.vB_rd[0][kk])))
        for i in range(self.nagpr):
            e("v_accvgpr_write_b32 a%d, 0" % i)
        e("s_waitcnt vmcnt(%d)" % (self.NPA + (c.NB - 1) * c.NPB), "weight stage 0 and tile 0 have landed")
        e("s_barrier")
        for ins in self.frag_reads(0, 0, 0, 0):
            e(ins)
        e("s_mov_b32 %s, %s" % (R("s", self.s_cnt), R("s", self.s_nch)))

    def adv_a(self):
        e = self.e
        e("s_add_u32 %s, %s, 128" % (R("s", self.s_cA), R("s", self.s_cA)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", self.s_clast), R("s", self.s_srcAw)))
        e("s_min_u32 %s, %s, %s" % (R("s", self.s_cA), R("s", self.s_cA), R("s", self.s_t0)))

    def adv_b(self):
        e = self.e
        e("s_add_u32 %s, %s, 128" % (R("s", self.s_cB), R("s", self.s_cB)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_t0), R("s", self.s_clast), R("s", self.s_srcBw)))
        e("s_min_u32 %s, %s, %s" % (R("s", self.s_cB), R("s", self.s_cB), R("s", self.s_t0)))

    def adv_insts(self, s, w):
        return ["s_add_u32 %s, %s, 128" % (R("s", s), R("s", s)),
                "s_add_u32 %s, %s, %s" % (R("s", self.s_stg), R("s", self.s_clast), R("s", w)),
                "s_min_u32 %s, %s, %s" % (R("s", s), R("s", s), R("s", self.s_stg))]

    def mainloop(self):
        c, e = self.c, self.e
        from math import gcd
        trip = c.NA * c.NB // gcd(c.NA, c.NB)
        self.comment("---- main loop: %d chunks per trip (activation buffer = chunk %% %d, weight stage = chunk %% %d), 2 substeps of %d MFMAs" % (trip, c.NA, c.NB, c.MFR * c.NT))
        top, done = self.newlabel("loop"), self.newlabel("done")
        self.label(top)
        for ch in range(trip):
            a, b = ch % c.NA, ch % c.NB
            a1, b1 = (ch + 1) % c.NA, (ch + 1) % c.NB
            a2 = (ch + 2) % c.NA
            # ---- substep 0: compute on set 0, read (chunk, kk 1) into set 1, request the activation tile of chunk + 2
            self.comment("chunk %d substep 0" % ch)
            e("s_waitcnt lgkmcnt(0)")
            groups = [[r] for r in self.frag_reads(1, 1, a, b)]
            pieces = [self.a_piece(k, a2, self.s_cA) for k in range(self.NPA)]
            pieces[-1] = pieces[-1] + self.adv_insts(self.s_cA, self.s_srcAw)
            if c.probe & 1:
                pieces = []
            self.interleave(self.mfmas(0), self.merge(groups, pieces))
            # ---- substep 1: the stage barrier (chunk + 1 has landed for every wave; this chunk's fragments are all read), compute on
            # set 1, read (chunk + 1, kk 0), request the weight slab of chunk + NB into the stage just released
            self.comment("chunk %d substep 1" % ch)
            e("s_waitcnt vmcnt(%d)" % (self.NPA + (c.NB - 2) * c.NPB), "in flight across the barrier: the tile of chunk + 2 (and a weight slab when the ring has 3 stages)")
            e("s_waitcnt lgkmcnt(0)")
            if not c.probe & 4:
                e("s_barrier")
            groups = [[r] for r in self.frag_reads(0, 0, a1, b1)]
            pieces = [self.b_piece(i, b, self.s_cB) for i in range(c.NPB)]
            pieces[-1] = pieces[-1] + self.adv_insts(self.s_cB, self.s_srcBw)
            if c.probe & 1:
                pieces = []
            self.interleave(self.mfmas(1), self.merge(groups, pieces))
            e("s_sub_u32 %s, %s, 1" % (R("s", self.s_cnt), R("s", self.s_cnt)))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_cnt))
            if ch < trip - 1:
                e("s_cbranch_scc1 %s" % done)
            else:
                e("s_cbranch_scc0 %s" % top)
        self.label(done)

    def finish(self):
        c = self.c
        text = super().finish()
        return text


def _variants():
    v = {}
    fam = (
            # layer 3 (14 x 14, tile = one image): conv1 forward 1024 -> 256 with BN statistics (and without), conv3 data gradient with the BN-backward sums
            (1024, 256, 196, 2, (0, 1, 2)),
            # layer 4 (7 x 7, tile = two images): 2048 -> 512
            (2048, 512, 98, 3, (0, 1, 2)),
            # conv1 of the first blocks (it runs before the stride): layer 3 512 -> 256 at 28 x 28, layer 4 1024 -> 512 at 14 x 14
            (512, 256, 196, 2, (0, 1)),
            (1024, 512, 196, 2, (0, 1)),
            # layer 4 conv3 forward 512 -> 2048 (tile = four images)
            # (not shipped: 256 -> 1024, layer 3 conv3 forward — 4 chunks per workgroup; 57-61 us against 47.7 us of pw_gen.py's resident-K kernel)
            (512, 2048, 196, 2, (0, 1)),
            # layer 2 (28 x 28): conv1 forward 512 -> 128 and conv3 data gradient: 128-column tiles (two 16-column tiles per wave)
            (512, 128, 196, 2, (0, 1, 2), 2))
    # (not shipped: 256 -> 128 at 56 x 56, layer 2's first conv1 — 173 us against 156 us of the implicit-GEMM kernel, profiles/r05_*)
    # the same families with tiles of 200 / 100 pixels: the pixel counts of the 160 px and 320 px stages of the progressive-resize recipe
    # (BASELINE configs[4]: 10 x 10 and 20 x 20 at layer 3, 5 x 5 and 10 x 10 at layer 4) are multiples of 100, not of 49
    # BASELINE configs[3] (BResNet-50): conv3's data gradient of layers 3 / 4 with bn2's backward sums under the leaky mask (stats 3, dconv_gen.py)
    fam = fam + ((1024, 256, 196, 2, (3,)), (2048, 512, 98, 3, (3,)))
    for (K, N, W, NB, stats, *rest) in fam + tuple((K, N, {196: 200, 98: 100}[W], NB, st, *rest) for (K, N, W, NB, st, *rest) in fam):
        for st in stats:
            name = "pk_k%d_n%d_w%d_s%d" % (K, N, W, st)
            v[name] = mk(name, W, K, N, st, NB=NB, NT=rest[0] if rest else 4)
    return v


VARIANTS = _variants()


def generate(name, **over):
    c = VARIANTS[name]
    if over:
        c = PkCfg(**{**c.__dict__, **over})
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
