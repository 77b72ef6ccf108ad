#!/usr/bin/env python3
"""wg1_gen.py — generator of the hand-scheduled gfx950 (MI355X) weight-gradient kernels of the 1x1 convolutions.

What the kernels replace: the cuDNN weight-gradient under `loss.backward()` of the reference (call form
/root/reference/sota_imagenet/callbacks.py:316-317) for the pointwise convolutions of ResNet-50's layers 3 and 4 — the WgradArgs
contract of conv_wgrad.hip (fp32 partial slabs [split][Cout][Cin], summed in a fixed order by splitk_reduce).

    dW[co][ci] = sum over pixels p of dy[p][co] * in[p][ci]

These launches are HBM-bound (26 GFLOP over 129 MB at layer 3: 14 us of MFMA against 23 us of HBM), so the structure is built around
keeping bytes in flight, not around the MFMA pipe:
  tile        128 input channels x 256 output channels per workgroup (4 waves = one per SIMD; wave w owns output channels 64w .. 64w+63
              and all 128 input channels: 8 x 4 accumulator tiles of 16 x 16, 128 AGPRs); grid = splits x (ci tiles x co tiles): the
              workgroups of one split read the same pixels and share an XCD (workgroup id % 8), so every byte leaves HBM once and
              the re-reads of the other channel tiles hit that XCD's L2 (measured: 55 -> 27 us at layer 3 against the other order).
  reduction   flat pixel ranges: a tile is 64 consecutive pixels, staged as six [64 positions][128 B] planes (4 of dy, 2 of in) by
              LDS-DMA; THREE tile buffers: the pieces of tile t + 3 are issued behind the barrier of tile t and have two tile times to
              land (counted vmcnt: one tile's pieces stay in flight across the barrier).  No padding: the last tile of a tensor whose
              pixel count is not a multiple of 64 reads zeros past the descriptor.
  operands    ds_read_b64_tr_b16 (both operands are k-major in memory), the k permutation and the chunk swizzle of wg_gen.py.
  epilogue    accumulators straight from AGPRs to this split's fp32 slab.
"""
import argparse
import os
import sys
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dconv_gen import Alloc, R  # noqa: E402
import wg_gen  # noqa: E402


@dataclass
class W1Cfg:
    name: str
    C: int        # channels of the input tensor
    CO: int       # channels of dy
    XP: int = 2   # in planes of 64 channels per workgroup
    DP: int = 4   # dy planes of 64 channels per workgroup; a wave owns DP of the 4 * DP dy fragments (16 * DP output channels)
    NBUF: int = 3

    TP = 64       # positions per tile

    @property
    def BUF(self):
        return (self.XP + self.DP) * self.TP * 128

    @property
    def LDS(self):
        return self.NBUF * self.BUF

    @property
    def NCI(self):
        return self.C // (64 * self.XP)

    @property
    def NCO(self):
        return self.CO // (64 * self.DP)


class Gen(wg_gen.Gen):
    KA = dict(dy=0, x=8, partial=16, tps=24, ntiles=28, npix=32, size=64)

    def __init__(self, c):
        self.c = c
        self.out = []
        self.nlabel = 0
        self.S = Alloc("s", 4, 100)
        self.V = Alloc("v", 1, 256)


    def gen(self):
        c, S, V = self.c, self.S, self.V
        self.NPC = 2 * (c.XP + c.DP)   # pieces per wave and tile: every plane's two 8-position blocks w and w + 4
        self.NF, self.NX = c.DP, 4 * c.XP  # dy / in fragments of a wave
        self.s_split, self.s_pair = 2, 3   # grid = (splits, channel-tile pairs): the workgroups of one split share their pixels' bytes and
        # land on the same XCD (workgroup id % 8), so the sharing happens in that XCD's L2
        self.srd = {"d": S.get(4, 4), "x": S.get(4, 4)}
        self.srdP = S.get(4, 4)
        self.s_ka = S.get(12, 4)
        (self.s_w, self.s_tile, self.s_tend, self.s_cnt, self.s_t0, self.s_t1, self.s_t2, self.s_t3, self.s_ci, self.s_co, self.s_ldsw,
         self.s_outw) = [S.get() for _ in range(12)]
        self.s_srcw = {"d": S.get(), "x": S.get()}
        self.s_tsrc = {"d": S.get(), "x": S.get()}
        # read bases per buffer: dy fragment n of this wave's plane, in fragment n of plane 0 (plane 1: + 8192 as an immediate)
        self.vD_rd = [[V.get() for n in range(self.NF)] for b in range(c.NBUF)]
        self.vX_rd = [[V.get() for n in range(4)] for b in range(c.NBUF)]
        self.v_dma = {"d": V.get(), "x": V.get()}
        self.v_tmp = [V.get(), V.get()]
        self.v_out = V.get()
        self.F = []
        for s in range(2):
            fd = V.get(4 * self.NF, 4)
            fx = V.get(4 * self.NX, 4)
            self.F.append((fd, fx))
        self.v_t = [self.F[1][1] + i for i in range(12)]
        self.nvgpr = V.n
        self.accum_offset = (self.nvgpr + 7) // 8 * 8
        self.nagpr = 4 * self.NF * self.NX
        self.tmp_i = 0
        self.prologue()
        self.mainloop()
        self.epilogue()
        return self.finish()

    # -----------------------------------------------------------------------------------------------------------------
    def tile_setup(self):
        """source offsets of tile s_tile: pixel 64 * tile"""
        c = self.c
        out = []
        for k, ch in (("d", c.CO), ("x", c.C)):
            out.append("s_mul_i32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_tile), c.TP * ch * 2))
            out.append("s_add_u32 %s, %s, %s" % (R("s", self.s_tsrc[k]), R("s", self.s_srcw[k]), R("s", self.s_t0)))
        return out

    def piece_insts(self, j, buf):
        """piece j of this wave: plane j >> 1 (0..3 dy, 4..5 in), 8-position block w + 4 (j & 1)"""
        c = self.c
        p, hb = j >> 1, j & 1
        k = "d" if p < c.DP else "x"
        ch = c.CO if k == "d" else c.C
        pl = p if k == "d" else p - c.DP
        vt = self.v_tmp[self.tmp_i & 1]
        self.tmp_i += 1
        return ["s_add_u32 m0, %s, %d" % (R("s", self.s_ldsw), buf * c.BUF + p * c.TP * 128 + hb * 4096),
                "s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_tsrc[k]), hb * 32 * ch * 2 + pl * 128),
                "v_add_u32 %s, %s, %s" % (R("v", vt), R("s", self.s_t0), R("v", self.v_dma[k])),
                "buffer_load_dwordx4 %s, %s, 0 offen lds" % (R("v", vt), R("s", self.srd[k], 4))]

    def all_pieces(self, buf):
        return [self.piece_insts(j, buf) for j in range(self.NPC)]

    def frag_reads(self, fset, step, buf):
        c = self.c
        fd, fx = self.F[fset]
        out = []
        for n in range(self.NF):
            for h in range(2):
                out.append("ds_read_b64_tr_b16 %s, %s offset:%d" % (R("v", fd + 4 * n + 2 * h, 2), R("v", self.vD_rd[buf][n]), step * 4096 + h * 2048))
        for j in range(self.NX):
            q, n = divmod(j, 4)
            for h in range(2):
                out.append("ds_read_b64_tr_b16 %s, %s offset:%d" % (R("v", fx + 4 * j + 2 * h, 2), R("v", self.vX_rd[buf][n]), q * 8192 + step * 4096 + h * 2048))
        return out

    def mfmas(self, fset):
        fd, fx = self.F[fset]
        out = []
        for j in range(self.NX):
            for n in range(self.NF):
                acc = (j * self.NF + n) * 4
                out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (R("a", acc, 4), R("v", fx + 4 * j, 4), R("v", fd + 4 * n, 4), R("a", acc, 4)))
        return out

    # -----------------------------------------------------------------------------------------------------------------
    def prologue(self):
        c, e = self.c, self.e
        ka = self.s_ka
        v = self.v_t
        t0, t1 = self.s_t0, self.s_t1
        self.comment("---- prologue")
        e("s_load_dwordx8 %s, s[0:1], 0x0" % R("s", ka, 8))     # dy, x, partial, tps, ntiles
        e("s_load_dword %s, s[0:1], 0x20" % R("s", ka + 8))      # npix
        lane = v[0]
        e("v_lshrrev_b32 %s, 6, v0" % R("v", v[1]))
        e("v_and_b32 %s, 63, v0" % R("v", lane))
        e("v_readfirstlane_b32 %s, %s" % (R("s", self.s_w), R("v", v[1])))
        # ---- lane parts of the LDS-DMA pieces: position-in-block = lane >> 3, chunk = (lane & 7) ^ (row & 6)
        l3, l7, ch16, off = v[3], v[4], v[5], v[7]
        e("v_lshrrev_b32 %s, 3, %s" % (R("v", l3), R("v", lane)))
        e("v_and_b32 %s, 7, %s" % (R("v", l7), R("v", lane)))
        e("v_and_b32 %s, 6, %s" % (R("v", ch16), R("v", l3)))
        e("v_xor_b32 %s, %s, %s" % (R("v", ch16), R("v", l7), R("v", ch16)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", ch16), R("v", ch16)))
        for k, chn in (("d", c.CO), ("x", c.C)):
            e("v_mov_b32 %s, %d" % (R("v", off), chn * 2))
            e("v_mad_u32_u24 %s, %s, %s, %s" % (R("v", self.v_dma[k]), R("v", l3), R("v", off), R("v", ch16)))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_srcw[k]), R("s", self.s_w), 8 * chn * 2), "this wave's 8-position block of a 32-position half")
        e("s_lshl_b32 %s, %s, 10" % (R("s", self.s_ldsw), R("s", self.s_w)))
        e("s_waitcnt lgkmcnt(0)")
        # ---- tile pair, split, descriptors
        nci = c.NCI
        assert nci & (nci - 1) == 0
        e("s_and_b32 %s, %s, %d" % (R("s", self.s_ci), R("s", self.s_pair), nci - 1))
        e("s_lshr_b32 %s, %s, %d" % (R("s", self.s_co), R("s", self.s_pair), nci.bit_length() - 1))
        nslab = c.CO * c.C * 4
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_split), nslab))
        e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", self.s_split), nslab))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t2), R("s", self.s_co), c.DP * 64 * c.C * 4))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_t3), R("s", self.s_ci), c.XP * 64 * 4))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_t2), R("s", self.s_t2), R("s", self.s_t3)))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", self.s_t2)))
        e("s_addc_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdP), R("s", ka + 4), R("s", t0)))
        e("s_addc_u32 %s, %s, %s" % (R("s", self.srdP + 1), R("s", ka + 5), R("s", t1)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdP + 1), R("s", self.srdP + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdP + 2), c.DP * 64 * c.C * 4))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdP + 3))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_outw), R("s", self.s_w), 16 * c.DP * c.C * 4), "this wave's output channels in the slab")
        # channel tile byte offsets into the pixel rows
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_ci), R("s", self.s_ci), c.XP * 128))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_co), R("s", self.s_co), c.DP * 128))
        for k, chn, ptr, tile_ch in (("d", c.CO, 0, self.s_co), ("x", c.C, 2, self.s_ci)):
            srd = self.srd[k]
            e("s_add_u32 %s, %s, %s" % (R("s", srd), R("s", ka + ptr), R("s", tile_ch)))
            e("s_addc_u32 %s, %s, 0" % (R("s", srd + 1), R("s", ka + ptr + 1)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", srd + 1), R("s", srd + 1)))
            e("s_mul_i32 %s, %s, %d" % (R("s", srd + 2), R("s", ka + 8), chn * 2), "tensor bytes (< 4 GiB)")
            e("s_sub_u32 %s, %s, %s" % (R("s", srd + 2), R("s", srd + 2), R("s", tile_ch)))
            e("s_mov_b32 %s, 0x00020000" % R("s", srd + 3))
        e("s_mul_i32 %s, %s, %s" % (R("s", self.s_tile), R("s", self.s_split), R("s", ka + 6)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_tend), R("s", self.s_tile), R("s", ka + 6)))
        e("s_min_u32 %s, %s, %s" % (R("s", self.s_tend), R("s", self.s_tend), R("s", ka + 7)))
        e("s_sub_u32 %s, %s, %s" % (R("s", self.s_cnt), R("s", self.s_tend), R("s", self.s_tile)), "tiles of this workgroup (>= 1)")
        e("s_sub_u32 %s, %s, 1" % (R("s", self.s_tend), R("s", self.s_tend)))
        # ---- first loads: tiles 0, 1, 2 into buffers 0, 1, 2
        for b in range(c.NBUF):
            if b:
                for ins in self.next_tile_insts():
                    e(ins)
            for ins in self.tile_setup():
                e(ins)
            ps = self.all_pieces(b)
            for grp in (ps if b < c.NBUF - 1 else ps[:len(ps) // 2]):   # (the last tile's second half: first k-step of the main loop)
                for ins in grp:
                    e(ins)
        # ---- lane offset of the slab stores: ((lane & 15) * C + 4 (lane >> 4)) * 4
        e("v_and_b32 %s, 15, %s" % (R("v", v[8]), R("v", lane)))
        e("v_lshrrev_b32 %s, 4, %s" % (R("v", v[9]), R("v", lane)))
        e("v_mov_b32 %s, %d" % (R("v", v[10]), c.C * 4))
        e("v_mul_lo_u32 %s, %s, %s" % (R("v", v[8]), R("v", v[8]), R("v", v[10])))
        e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", self.v_out), R("v", v[9]), R("v", v[8])))
        # ---- transposed-read bases: row = lane >> 2 (= 4g + q), p = lane & 3; addr = row*128 + ((2n ^ (row & 6)) * 16) + 8p
        row, p8, cc, rk = v[3], v[4], v[5], v[6]
        e("v_lshrrev_b32 %s, 2, %s" % (R("v", row), R("v", lane)))
        e("v_and_b32 %s, 3, %s" % (R("v", p8), R("v", lane)))
        e("v_lshlrev_b32 %s, 3, %s" % (R("v", p8), R("v", p8)))
        e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", p8), R("v", row), R("v", p8)))
        e("v_and_b32 %s, 6, %s" % (R("v", cc), R("v", row)))
        for n in range(4):
            e("v_xor_b32 %s, %d, %s" % (R("v", rk), 2 * n, R("v", cc)))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", rk), R("v", rk), R("v", p8)))
            for b in range(c.NBUF):
                e("v_add_u32 %s, %d, %s" % (R("v", self.vX_rd[b][n]), b * c.BUF + c.DP * c.TP * 128, R("v", rk)))
        # dy fragment f of this wave = fragment (w * DP + f) of the workgroup's 4 * DP: plane (.. >> 2), fragment-in-plane (.. & 3)
        for f in range(self.NF):
            e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_w), c.DP))
            e("s_add_u32 %s, %s, %d" % (R("s", t0), R("s", t0), f))
            e("s_and_b32 %s, %s, 3" % (R("s", t1), R("s", t0)))
            e("s_lshl_b32 %s, %s, 1" % (R("s", t1), R("s", t1)))
            e("s_lshr_b32 %s, %s, 2" % (R("s", t0), R("s", t0)))
            e("s_lshl_b32 %s, %s, 13" % (R("s", t0), R("s", t0)))
            e("v_xor_b32 %s, %s, %s" % (R("v", rk), R("s", t1), R("v", cc)))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", rk), R("v", rk), R("v", p8)))
            e("v_add_u32 %s, %s, %s" % (R("v", rk), R("s", t0), R("v", rk)))
            for b in range(c.NBUF):
                e("v_add_u32 %s, %d, %s" % (R("v", self.vD_rd[b][f]), b * c.BUF, R("v", rk)))
        for i in range(self.nagpr):
            e("v_accvgpr_write_b32 a%d, 0" % i)
        e("s_waitcnt vmcnt(%d)" % (self.NPC + self.NPC // 2), "tile 0 has landed")
        e("s_barrier")
        for ins in self.frag_reads(0, 0, 0):
            e(ins)

    def mainloop(self):
        c, e = self.c, self.e
        self.comment("---- main loop: %d tiles per trip (the LDS buffers rotate), 2 k-steps of %d MFMAs per tile" % (c.NBUF, self.NF * self.NX))
        top, done = self.newlabel("loop"), self.newlabel("done")
        self.label(top)
        for b in range(c.NBUF):
            nb = (b + 1) % c.NBUF
            # ---- k-step 0: compute on set 0, read step 1 into set 1; the second half of the pieces issued behind the previous barrier
            self.comment("buffer %d k-step 0" % b)
            e("s_waitcnt lgkmcnt(0)")
            groups = [[r] for r in self.frag_reads(1, 1, b)]
            pb = (b + c.NBUF - 1) % c.NBUF     # the buffer the previous tile freed
            ps = self.all_pieces(pb)
            half = len(ps) // 2
            self.interleave(self.mfmas(0), self.merge(groups, ps[half:]))
            # ---- k-step 1: barrier (the next tile has landed; this buffer's fragments are all read), compute on set 1, read the next
            # tile's step 0, first half of the pieces of tile + NBUF into this buffer
            self.comment("buffer %d k-step 1" % b)
            e("s_waitcnt lgkmcnt(0)")
            # outstanding and allowed to stay in flight: the pieces of tile + 2 (issued behind the previous barrier and in step 0)
            e("s_waitcnt vmcnt(%d)" % self.NPC)
            e("s_barrier")
            groups = [[r] for r in self.frag_reads(0, 0, nb)]
            ps = self.all_pieces(b)
            setup = self.next_tile_insts() + self.tile_setup()
            first = [setup + ps[0]] + ps[1:half]
            self.interleave(self.mfmas(1), self.merge(groups, first))
            e("s_sub_u32 %s, %s, 1" % (R("s", self.s_cnt), R("s", self.s_cnt)))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_cnt))
            if b < c.NBUF - 1:
                e("s_cbranch_scc1 %s" % done)
            else:
                e("s_cbranch_scc0 %s" % top)
        self.label(done)

    def epilogue(self):
        c, e = self.c, self.e
        self.comment("---- epilogue: the accumulator tiles -> this split's slab")
        e("s_waitcnt vmcnt(0)", "the look-ahead pieces (never used) have landed: no LDS-DMA is in flight when the wave ends")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 15")
        for j in range(self.NX):
            for n in range(self.NF):
                acc = (j * self.NF + n) * 4
                e("s_add_u32 %s, %s, %d" % (R("s", self.s_t0), R("s", self.s_outw), (n * 16 * c.C + j * 16) * 4))
                e("buffer_store_dwordx4 %s, %s, %s, %s offen" % (R("a", acc, 4), R("v", self.v_out), R("s", self.srdP, 4), R("s", self.s_t0)))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")


# ---------------------------------------------------------------------------------------------------------------------
VARIANTS = {
    # ResNet-50: conv1 / conv3 of the bottlenecks of layer 3 (1024 <-> 256 channels) and layer 4 (2048 <-> 512), any pixel count
    "wg1_c1024_o256": W1Cfg("wg1_c1024_o256", C=1024, CO=256),
    "wg1_c256_o1024": W1Cfg("wg1_c256_o1024", C=256, CO=1024),
    "wg1_c2048_o512": W1Cfg("wg1_c2048_o512", C=2048, CO=512),
    "wg1_c512_o2048": W1Cfg("wg1_c512_o2048", C=512, CO=2048),
    # first blocks of layers 3 and 4 (conv1 runs before the stride), layer 2 (512 <-> 128, 256 -> 128), layer 1 (256 <-> 64)
    "wg1_c512_o256": W1Cfg("wg1_c512_o256", C=512, CO=256),
    "wg1_c1024_o512": W1Cfg("wg1_c1024_o512", C=1024, CO=512),
    "wg1_c512_o128": W1Cfg("wg1_c512_o128", C=512, CO=128, XP=4, DP=2),
    "wg1_c64_o256": W1Cfg("wg1_c64_o256", C=64, CO=256, XP=1, DP=4),
    # measured in the executor and NOT shipped (the implicit-GEMM kernel is as fast or faster there; all are HBM-bound at ~3.8 TB/s):
    #   C=128 -> CO=512 (layer 2 conv3): 53.1 vs 51.5 us;  C=256 -> CO=128 (XP=4, DP=2; layer 2 block 0 conv1): 142 vs 136 us;
    #   C=256 -> CO=64 (XP=4, DP=1; layer 1 conv1): 106.8 vs 95.8 us.  The generator still builds them (tests run the DP=1 form).
}


EXTRA = {"wg1_c256_o64": W1Cfg("wg1_c256_o64", C=256, CO=64, XP=4, DP=1)}  # test-only tile shape (see above)


def generate(name, **over):
    c = VARIANTS.get(name) or EXTRA[name]
    if over:
        c = W1Cfg(**{**c.__dict__, **over})
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
