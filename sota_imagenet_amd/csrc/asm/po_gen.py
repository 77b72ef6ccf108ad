#!/usr/bin/env python3
"""po_gen.py — generator of the hand-scheduled gfx950 (MI355X) pointwise convolution kernels for the OUTPUT-HEAVY, HBM-bound 1x1
launches of ResNet-50's train step: a short reduction K (64 .. 512 channels) into 4K columns under a streaming epilogue —
  * conv3 forward / the stride-1 downsample forward (BN statistics rows of the output),
  * conv1's data gradient: + the shortcut gradient (addend, under its ReLU bit mask or plain) + the BN-backward sums of the layer
    whose activation gradient the output is (that layer's y and ReLU bit mask are read in the epilogue) —
the IgemmArgs contract of launch_igemm() (conv forward / data gradient under `model(data)` / `loss.backward()`,
/root/reference/sota_imagenet/callbacks.py:316-317).

Why its own structure.  These launches move 3 .. 13 bytes of epilogue traffic per byte of GEMM input and carry 10 us of MFMA work
against 60 .. 250 us of HBM time: what bounds them is bytes in flight per CU and the epilogue's instruction count, not the matrix
pipe.  So:
  weights     RESIDENT IN AGPRs.  Waves are 1 (M) x 4 (N); a wave owns BN/4 columns, and its whole K x BN/4 weight slab (8 .. 32 KiB)
              is loaded ONCE per workgroup straight into accumulation registers (MFMA reads its weight operand from AGPRs): no weight
              stream, no weight ring, no LDS traffic for weights for the rest of the kernel.
  workgroups  persistent: one per CU, a workgroup owns ONE column tile and a contiguous run of 64-pixel tiles; the workgroups of the
              column tiles of one pixel run sit on one XCD (workgroup id % 8), so the pixel tile and the 128-byte mask lines they
              share are fetched from HBM once.
  A operand   the 64 x K pixel tile by LDS-DMA ([plane of 64 channels][pixel][128 B], XOR-swizzled chunks), two buffers: tile t + 1
              is requested right behind the barrier of tile t.
  epilogue    straight from the accumulators, one (fragment, tile pair) ITEM at a time: 8 consecutive channels of one pixel per lane;
              the y / mask / addend vectors of an item live in 10 registers that are re-requested for the NEXT tile the moment the
              item has consumed them (rolling refill): a full tile's worth of epilogue operands (20 KiB per wave) is in flight at
              every moment, and every wait is a counted vmcnt with a constant derived below.
  statistics  per-lane sums stay in registers across ALL tiles of the workgroup; one DPP row reduction and one partial row per
              workgroup at the end.
Tensor extents are enforced by the buffer descriptors (num_records = bytes left from the tile's base, moved tile by tile), so a
ragged last tile (pixel count not a multiple of 64) reads zeros and drops its stores; no SGPR offset takes part in a range check.
"""
import argparse
import os
import sys
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dconv_gen import Alloc, R, LEAKY_BITS  # noqa: E402


@dataclass
class PoCfg:
    name: str
    K: int            # input channels (reduction), multiple of 64
    BN: int           # output columns per workgroup (256 | 128)
    stats: int        # 0 none, 1 BN statistics of the output, 2 BN-backward sums, 3 BN-backward sums under a leaky-ReLU mask (slope 0.01: dconv_gen.py Cfg.stats)
    add: int = 0      # 0 no addend, 1 addend, 2 addend under its ReLU bit mask, 3 addend given at HALF resolution: a compact [N][H/2][W/2][cols]
                      # tensor that stands for a full-resolution one whose odd rows / columns are zero (the data gradient of a stride-2 1x1
                      # convolution — the downsample branch of a stage's first block — which is then never written at full size)
    MFR: int = 4      # 16-pixel fragments per tile
    WM: int = 1       # waves along the pixel dimension: 1 -> waves 1 (M) x 4 (N), every wave computes all MFR fragments of its BN/4 columns;
                      # 2 -> waves 2 x 2: a wave computes MFR/2 fragments of BN/2 columns (the 64-column launches of layer 1: a wave still owns a
                      # tile pair = 8 consecutive channels per lane; two partial rows per workgroup)
    NBUF: int = 2     # A tile buffers
    tmask: int = 1    # 1: the ReLU mask bytes of a whole 64-pixel tile by ONE load per mask tensor (lane = pixel, 8 or 4 bytes = this wave's columns),
                      #    handed to the lanes that need them by ds_bpermute; 0: one byte load per (fragment, tile pair) and lane
    weave: int = 1    # 1: two accumulator sets, the MFMAs of tile t + 1 issued between the epilogue instructions of tile t
    nt: int = 0       # non-temporal cache policy: 1 epilogue operand loads, 2 output stores, 4 A pieces
    probe: int = 0    # timing probes (WRONG results): 1 no MFMAs, 2 no epilogue arithmetic, 4 no operand loads, 8 no stores
    bnin: int = 0     # 1: the input is the RAW output y of the previous convolution: a = relu(y * scale[c] + shift[c]) rounded to bf16 (bn_apply_kernel's
                      # value) is formed in LDS — every wave transforms the pieces its own LDS-DMA lanes wrote, between their landing and the tile's
                      # barrier — and the workgroups of column tile 0 leave a and its ReLU bits in memory (conv3's forward: bn2 + ReLU in the operand
                      # path; these launches are HBM-bound, the transform's ~40 VALU instructions per KiB hide under the memory time).  Pixel counts that
                      # are a multiple of the tile only (a ragged tile's missing pixels would become relu(shift), not zero)

    @property
    def WN(self):     # waves along the output columns
        return 4 // self.WM

    @property
    def MFRW(self):   # fragments per wave
        return self.MFR // self.WM

    @property
    def NT(self):     # 16-column tiles per wave
        return self.BN // (16 * self.WN)

    @property
    def KS(self):     # 32-channel k-steps
        return self.K // 32

    @property
    def NPL(self):    # planes of 64 channels
        return self.K // 64

    @property
    def TP(self):     # pixels per tile
        return self.MFR * 16

    @property
    def PLANE(self):
        return self.TP * 128

    @property
    def ABUF(self):
        return self.NPL * self.PLANE

    @property
    def LDS(self):
        return self.NBUF * self.ABUF

    @property
    def NPW(self):    # LDS-DMA pieces per wave and tile
        return self.NPL * self.TP // 8 // 4

    @property
    def NI(self):     # epilogue items per tile: (fragment, tile pair)
        return self.MFRW * self.NT // 2

    @property
    def NM(self):     # mask tensors (BN layer's ReLU bits, addend's ReLU bits)
        return (1 if self.stats >= 2 else 0) + (1 if self.add == 2 else 0)

    @property
    def L(self):      # vector-memory loads per item
        return (1 if self.stats >= 2 else 0) + (1 if self.add else 0) + (0 if self.tmask else self.NM)

    @property
    def NMT(self):    # mask loads per tile (tmask)
        return self.NM if self.tmask else 0

    @property
    def FULL(self):   # full-line stores: a wave owns 128 bytes of every pixel row (two tile pairs), a store instruction writes 8 pixels x 128 bytes
        return self.NT == 4


class Gen:
    KA = dict(in_=0, wt=8, out=16, stat=24, bn_y=32, bn_bits=40, bn_mean=48, bn_invstd=56, addend=64, addend_bits=72,
              npix=80, ncols=84, tpg=88, ngroups=92, ntiles=96, lognct=100, W=104, H=108, magic_w=112, magic_h=116, size=128)

    def __init__(self, c: PoCfg):
        self.c = c
        self.out = []
        self.nlabel = 0
        self.S = Alloc("s", 4, 102)
        self.V = Alloc("v", 1, 256)

    def e(self, s, comment=None):
        self.out.append("\t" + s + ("\t; " + comment if comment else ""))

    def label(self, name):
        self.out.append(name + ":")

    def newlabel(self, stem):
        self.nlabel += 1
        return "L_%s_%s_%d" % (self.c.name, stem, self.nlabel)

    def comment(self, s):
        self.out.append("\t; " + s)

    # -----------------------------------------------------------------------------------------------------------------
    def gen(self):
        c, S, V = self.c, self.S, self.V
        assert c.BN in (64, 128, 256) and c.K % 64 == 0 and c.MFR % 2 == 0 and c.NT % 2 == 0 and c.WM in (1, 2) and c.MFR % c.WM == 0
        assert c.WM == 1 or (not c.FULL and (not c.NMT or c.MFRW == 4)), "2 x 2 waves: half-line stores; the tile-wide mask load is one lane per pixel of the wave's 64"
        assert (c.L + 1) * c.NI + c.NPW + c.NMT <= 63, "vmcnt range"
        assert c.LDS <= 160 * 1024
        assert not c.bnin or (c.stats == 1 and c.add == 0 and c.WM == 1 and c.weave and c.NBUF == 2)
        self.s_wg = 2
        self.srdA, self.srdB, self.srdO, self.srdX = S.get(4, 4), S.get(4, 4), S.get(4, 4), S.get(4, 4)
        self.srdY = self.srdM = self.srdAD = self.srdAB = None
        if c.stats >= 2:
            self.srdY, self.srdM = S.get(4, 4), S.get(4, 4)
        if c.add:
            self.srdAD = S.get(4, 4)
        if c.add == 2:
            self.srdAB = S.get(4, 4)
        self.s_ka = S.get(20, 4)
        self.s_kp = S.get(8, 4)
        (self.s_w, self.s_t0, self.s_t1, self.s_t2, self.s_t3, self.s_cnt, self.s_pf, self.s_tout, self.s_tbits, self.s_g, self.s_ct, self.s_ldsA, self.s_n4,
         self.s_ia, self.s_io, self.s_ib, self.s_8rows, self.s_pfa) = [S.get() for _ in range(18)]
        self.s_lo8 = S.get(2, 2)   # lanes 0 .. 7 of every row of 16
        if c.bnin:
            self.srdA2, self.srdBt, self.srdSS = S.get(4, 4), S.get(4, 4), S.get(4, 4)   # a out, its ReLU bits out (the A descriptor's offsets / 16), [2][K] scale / shift
            self.s_trleft, self.s_trflag, self.s_k1, self.s_ta, self.s_tinc, self.s_tincb = [S.get() for _ in range(6)]
        self.s_wn, self.s_wm = (S.get(), S.get()) if c.WM > 1 else (self.s_w, None)   # this wave's column / pixel part
        self.vA_rd = [[V.get() for kk in range(2)] for b in range(c.NBUF)]
        self.vA_dma = V.get()
        self.v_tmp = [V.get(), V.get()]
        self.v_out_m = [V.get() for m in range(c.MFRW)]
        self.v_bits_m = [V.get() for m in range(c.MFRW)] if (c.NM and not c.tmask) else None
        if c.NMT:
            nd = c.NT // 2                       # dwords of a pixel's mask bytes owned by this wave (4 bytes per tile pair)
            self.v_mk = V.get()                  # lane = pixel: byte offset of its mask bytes
            self.v_bp = V.get()                  # (lane & 15) * 4: ds_bpermute address of fragment 0's pixel (+ 64 m by the offset field)
            self.v_kg8 = V.get()                 # (lane >> 4) * 8: this lane's byte of a tile pair's dword
            self.mk = {T: [V.get(nd, 2) for b in range(2)] for T in (["y"] if c.stats >= 2 else []) + (["a"] if c.add == 2 else [])}
            self.mkt = {T: [V.get() for p in range(nd)] for T in self.mk}   # the unit's mask dwords after the permute / shift
        self.v_chan = V.get()
        self.v_st_m = [V.get() for m in range(c.MFRW)] if c.FULL else None
        if c.add == 3:
            self.v_ad_m = [V.get() for m in range(c.MFRW)]   # this tile's half-resolution addend offsets (or out of range)
            self.v_r = V.get()                              # lane & 15
            self.v_col = V.get()                            # this lane's column bytes
            self.s_kq = S.get(4, 4)                         # magic_w, magic_h, -, -
            self.s_qpf = S.get()                            # first pixel of the prefetch tile
            self.s_vcc2 = S.get(2, 2)   # full-line stores: (m*16 + (r & 7)) rows, chunk (r >> 3)*64 + kg*16
        # per-item operand registers
        self.it = []
        for i in range(c.NI):
            d = {}
            if c.stats >= 2:
                d["y"] = V.get(4, 4)
            if c.add:
                d["ad"] = V.get(4, 4)
            if c.stats >= 2:
                d["yb"] = self.mkt["y"][i % (c.NT // 2)] if c.tmask else V.get()
            if c.add == 2:
                d["ab"] = self.mkt["a"][i % (c.NT // 2)] if c.tmask else V.get()
            self.it.append(d)
        npair = c.NT // 2
        self.s1 = [V.get(8, 4) for p in range(npair)] if c.stats else None
        self.s2 = [V.get(8, 4) for p in range(npair)] if c.stats else None
        self.F = [V.get(4 * max(c.MFRW, 3), 4) for s in range(2)]
        self.tv = V.get(8, 2)
        self.dsets = [V.get(4, 4) for _ in range(4)]
        self.v_xc = V.get(4, 4) if c.FULL else None   # exchange temporary
        self.xr = V.get(8, 2)
        self.yv = V.get(2, 2)
        self.v_m = V.get()
        if c.bnin:
            self.v_lane16 = V.get()
            self.v_ssoff = V.get()
            self.v_sc = [V.get(8, 4) for _ in range(c.NPL)]   # this lane's 8 channels of every 64-channel plane
            self.v_sh = [V.get(8, 4) for _ in range(c.NPL)]
            self.tr = [dict(ta=V.get(), d=V.get(4, 4), f=V.get(8, 4), bits=V.get(), o=V.get(), o2=V.get()) for _ in range(2)]
        self.nvgpr = V.n
        self.accum_offset = (self.nvgpr + 7) // 8 * 8
        self.aB = 0
        self.aACC = c.NT * c.KS * 4
        self.nacc = 2 if c.weave else 1
        self.accset = 0   # the set the epilogue being emitted reads
        self.nagpr = self.aACC + self.nacc * c.MFRW * c.NT * 4
        assert self.nagpr <= 256 and self.accum_offset + self.nagpr <= 512
        self.tmp_i = 0
        self.prologue()
        self.mainloop()
        self.finale()
        return self.finish()

    def acc(self, m, n, aset=None):
        aset = self.accset if aset is None else aset
        return self.aACC + aset * self.c.MFRW * self.c.NT * 4 + (m * self.c.NT + n) * 4

    def breg(self, n, ks):
        return self.aB + (n * self.c.KS + ks) * 4

    # -----------------------------------------------------------------------------------------------------------------
    def desc_from(self, srd, ptr_lo, off_lo, off_hi, total, comment=None):
        """srd = raw buffer at ptr + (off_hi:off_lo) with num_records = total - off_lo (total, off: SGPRs; total < 2^32 bytes)"""
        e = self.e
        e("s_add_u32 %s, %s, %s" % (R("s", srd), R("s", ptr_lo), R("s", off_lo)), comment)
        e("s_addc_u32 %s, %s, %s" % (R("s", srd + 1), R("s", ptr_lo + 1), R("s", off_hi) if off_hi is not None else "0"))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", srd + 1), R("s", srd + 1)))
        e("s_sub_u32 %s, %s, %s" % (R("s", srd + 2), R("s", total), R("s", off_lo)))
        e("s_mov_b32 %s, 0x00020000" % R("s", srd + 3))

    def desc_adv(self, srd, inc):
        return ["s_add_u32 %s, %s, %s" % (R("s", srd), R("s", srd), R("s", inc)),
                "s_addc_u32 %s, %s, 0" % (R("s", srd + 1), R("s", srd + 1)),
                "s_sub_u32 %s, %s, %s" % (R("s", srd + 2), R("s", srd + 2), R("s", inc))]

    def a_piece(self, j, buf):
        """A piece j of this wave: 8-pixel block 4 * (j % (RB / 4)) + w of plane j // (RB / 4), RB = blocks per plane"""
        c = self.c
        rb4 = c.TP // 8 // 4
        blk4, plane = j % rb4, j // rb4
        vt = self.v_tmp[self.tmp_i & 1]
        self.tmp_i += 1
        return ["s_add_u32 m0, %s, %d" % (R("s", self.s_ldsA), buf * c.ABUF + plane * c.PLANE + blk4 * 4096),
                "v_add_u32 %s, %d, %s" % (R("v", vt), blk4 * 32 * c.K * 2 + plane * 128, R("v", self.vA_dma)),
                "buffer_load_dwordx4 %s, %s, 0 offen%s lds" % (R("v", vt), R("s", self.srdA, 4), " nt" if c.nt & 4 else "")]

    def sub2_addr(self, m):
        """add == 3: v_ad_m[m] = byte offset of pixel q = (prefetch tile's first pixel) + 16 m + (lane & 15) in the half-resolution addend, or an
        out-of-range offset where the pixel has an odd row or column (the buffer then returns zeros: that pixel's addend is zero).
        q = (n*H + h)*W + w;  the addend's pixel is (n*(H/2) + h/2)*(W/2) + w/2.  Divisions by multiplication with floor(2^32 / d) + 1:
        exact while q * d < 2^32 (the host checks it)."""
        c = self.c
        if c.add != 3:
            return []
        t = [self.tv + i for i in range(6)]
        q, a, w, n, h, x = (R("v", r) for r in t)
        M, N, W, H = (R("s", self.s_kp + i) for i in (0, 1, 6, 7))
        mw, mh = R("s", self.s_kq), R("s", self.s_kq + 1)
        s0 = R("s", self.s_t3)
        return ["s_add_u32 %s, %s, %d" % (s0, R("s", self.s_qpf), 16 * m),
                "v_add_u32 %s, %s, %s" % (q, s0, R("v", self.v_r)),
                "v_mul_hi_u32 %s, %s, %s" % (a, q, mw),                      # a = q / W
                "v_mul_lo_u32 %s, %s, %s" % (x, a, W),
                "v_sub_u32 %s, %s, %s" % (w, q, x),                          # w = q % W
                "v_mul_hi_u32 %s, %s, %s" % (n, a, mh),                      # n = a / H
                "v_mul_lo_u32 %s, %s, %s" % (x, n, H),
                "v_sub_u32 %s, %s, %s" % (h, a, x),                          # h = a % H
                "v_or_b32 %s, %s, %s" % (x, w, h),
                "v_and_b32 %s, 1, %s" % (x, x),
                "v_cmp_eq_u32 vcc, 0, %s" % x,                               # both even
                "v_cmp_gt_u32 s[%d:%d], %s, %s" % (self.s_vcc2, self.s_vcc2 + 1, M, q),   # and a pixel of the tensor
                "s_and_b64 vcc, vcc, s[%d:%d]" % (self.s_vcc2, self.s_vcc2 + 1),
                "s_lshr_b32 %s, %s, 1" % (s0, H),
                "v_lshrrev_b32 %s, 1, %s" % (h, h),
                "v_mad_u32_u24 %s, %s, %s, %s" % (n, n, s0, h),              # n*(H/2) + h/2
                "s_lshr_b32 %s, %s, 1" % (s0, W),
                "v_lshrrev_b32 %s, 1, %s" % (w, w),
                "v_mad_u32_u24 %s, %s, %s, %s" % (n, n, s0, w),              # ... *(W/2) + w/2
                "s_lshl_b32 %s, %s, 1" % (s0, N),
                "v_mul_lo_u32 %s, %s, %s" % (n, n, s0),
                "v_add_u32 %s, %s, %s" % (n, n, R("v", self.v_col)),
                "v_mov_b32 %s, 0x80000000" % x,
                "v_cndmask_b32 %s, %s, %s, vcc" % (R("v", self.v_ad_m[m]), x, n)]

    def mask_loads(self, buf):
        """tmask: the mask bytes of the PREFETCH tile, lane = pixel, into mask buffer buf"""
        c = self.c
        if not c.NMT or (c.probe & 4):
            return []
        op = "buffer_load_dwordx2" if c.NT == 4 else "buffer_load_dword"
        out = []
        for T in self.mk:
            srd = self.srdM if T == "y" else self.srdAB
            out.append("%s %s, %s, %s, 0 offen" % (op, R("v", self.mk[T][buf], c.NT // 2), R("v", self.v_mk), R("s", srd, 4)) if not (c.probe & 16) else "s_nop 0")
        return out

    def mask_fetch(self, m, buf):
        """tmask: this lane's mask dwords of fragment m (pixel m*16 + (lane & 15) lives in lane m*16 + (lane & 15) of the tile's mask
        registers), shifted so that bit k of the low byte is the bit of this lane's element k"""
        c = self.c
        if not c.NMT or (c.probe & 4):
            return []
        out = []
        for T in self.mk:
            for p in range(c.NT // 2):
                out.append("ds_bpermute_b32 %s, %s, %s offset:%d" % (R("v", self.mkt[T][p]), R("v", self.v_bp), R("v", self.mk[T][buf] + p), 64 * m))
        out.append("s_waitcnt lgkmcnt(0)")
        for T in self.mk:
            for p in range(c.NT // 2):
                out.append("v_lshrrev_b32 %s, %s, %s" % (R("v", self.mkt[T][p]), R("v", self.v_kg8), R("v", self.mkt[T][p])))
        return out

    def item_loads(self, i):
        """the epilogue operand loads of item i = (fragment m, pair p) from the PREFETCH descriptors"""
        c = self.c
        if c.probe & 4:
            return []
        m, p = divmod(i, c.NT // 2)
        d = self.it[i]
        out = []
        nt = " nt" if c.nt & 1 else ""
        nomask, nobig = c.probe & 16, c.probe & 32   # (probes: the mask-byte loads / the 16-byte loads replaced by a scalar no-op each: same counts)
        if c.stats >= 2:
            out.append("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d%s" % (R("v", d["y"], 4), R("v", self.v_out_m[m]), R("s", self.srdY, 4), p * 64, nt) if not nobig else "s_nop 0")
            if not c.tmask:
                out.append("buffer_load_ubyte %s, %s, %s, 0 offen offset:%d%s" % (R("v", d["yb"]), R("v", self.v_bits_m[m]), R("s", self.srdM, 4), p * 4, nt) if not nomask else "s_nop 0")
        if c.add:
            va = self.v_ad_m[m] if c.add == 3 else self.v_out_m[m]
            out.append("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d%s" % (R("v", d["ad"], 4), R("v", va), R("s", self.srdAD, 4), p * 64, nt) if not nobig else "s_nop 0")
        if c.add == 2:
            if not c.tmask:
                out.append("buffer_load_ubyte %s, %s, %s, 0 offen offset:%d%s" % (R("v", d["ab"]), R("v", self.v_bits_m[m]), R("s", self.srdAB, 4), p * 4, nt) if not nomask else "s_nop 0")
        return out

    def frag_reads(self, fset, ks, buf):
        c = self.c
        plane, kk = ks // 2, ks % 2
        return ["ds_read_b128 %s, %s offset:%d" % (R("v", self.F[fset] + 4 * m, 4), R("v", self.vA_rd[buf][kk]), plane * c.PLANE + m * 2048) for m in range(c.MFRW)]

    # -----------------------------------------------------------------------------------------------------------------
    def prologue(self):
        c, e = self.c, self.e
        ka, kp = self.s_ka, self.s_kp
        t0, t1, t2, t3 = self.s_t0, self.s_t1, self.s_t2, self.s_t3
        v = [self.F[1] + i for i in range(12)]
        self.comment("---- prologue: kernel arguments, this workgroup's column tile and pixel run, descriptors, the weight slab into AGPRs")
        e("s_load_dwordx16 %s, s[0:1], 0x0" % R("s", ka, 16))
        e("s_load_dwordx4 %s, s[0:1], 0x40" % R("s", ka + 16, 4))
        e("s_load_dwordx8 %s, s[0:1], 0x50" % R("s", kp, 8), "npix, ncols, tpg, ngroups, ntiles, lognct, W, H")
        if c.add == 3:
            e("s_load_dwordx4 %s, s[0:1], 0x70" % R("s", self.s_kq, 4), "floor(2^32 / W) + 1, floor(2^32 / H) + 1")
        lane, r, kg = v[0], v[1], v[2]
        e("v_lshrrev_b32 %s, 6, v0" % R("v", v[3]))
        e("v_and_b32 %s, 63, v0" % R("v", lane))
        e("v_readfirstlane_b32 %s, %s" % (R("s", self.s_w), R("v", v[3])))
        e("v_and_b32 %s, 15, v0" % R("v", r))
        e("v_bfe_u32 %s, v0, 4, 2" % R("v", kg))
        if c.WM > 1:
            e("s_and_b32 %s, %s, %d" % (R("s", self.s_wn), R("s", self.s_w), c.WN - 1), "waves %d (M) x %d (N)" % (c.WM, c.WN))
            e("s_lshr_b32 %s, %s, %d" % (R("s", self.s_wm), R("s", self.s_w), c.WN.bit_length() - 1))
        e("s_waitcnt lgkmcnt(0)")
        M, N, TPG, G, T, LG = kp, kp + 1, kp + 2, kp + 3, kp + 4, kp + 5
        # group g (pixel run) and column tile ct: the NCT workgroups of a group are consecutive multiples of 8 apart -> same XCD
        e("s_and_b32 %s, %s, 7" % (R("s", t0), R("s", self.s_wg)), "XCD")
        e("s_lshr_b32 %s, %s, 3" % (R("s", t1), R("s", self.s_wg)))
        e("s_lshl_b32 %s, 1, %s" % (R("s", t2), R("s", LG)))
        e("s_sub_u32 %s, %s, 1" % (R("s", t2), R("s", t2)))
        e("s_and_b32 %s, %s, %s" % (R("s", self.s_ct), R("s", t1), R("s", t2)), "column tile")
        e("s_lshr_b32 %s, %s, %s" % (R("s", t1), R("s", t1), R("s", LG)))
        e("s_lshl_b32 %s, %s, 3" % (R("s", t1), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.s_g), R("s", t1), R("s", t0)), "group")
        e("s_cmp_ge_u32 %s, %s" % (R("s", self.s_g), R("s", G)))
        lab_run = self.newlabel("run")
        e("s_cbranch_scc0 %s" % lab_run)
        e("s_endpgm")
        self.label(lab_run)
        # tiles [g*tpg, min(g*tpg + tpg, T))
        e("s_mul_i32 %s, %s, %s" % (R("s", t0), R("s", self.s_g), R("s", TPG)), "first tile")
        e("s_add_u32 %s, %s, %s" % (R("s", t1), R("s", t0), R("s", TPG)))
        e("s_min_u32 %s, %s, %s" % (R("s", t1), R("s", t1), R("s", T)))
        e("s_sub_u32 %s, %s, %s" % (R("s", self.s_cnt), R("s", t1), R("s", t0)), "tiles of this workgroup (>= 1)")
        e("s_sub_u32 %s, %s, 1" % (R("s", self.s_pf), R("s", self.s_cnt)), "prefetch advances left")
        e("s_mov_b32 %s, %s" % (R("s", self.s_pfa), R("s", self.s_pf)), "... of the A descriptor (it runs one tile further ahead when the loop is woven)")
        self.s_first = S_first = self.S.get()
        e("s_mov_b32 %s, %s" % (R("s", S_first), R("s", t0)))
        e("s_mul_i32 %s, %s, %d" % (R("s", self.s_tout), R("s", N), c.TP * 2), "bytes of a tile of an output-shaped tensor")
        e("s_lshr_b32 %s, %s, 4" % (R("s", self.s_tbits), R("s", self.s_tout)))
        e("s_lshl_b32 %s, %s, 2" % (R("s", self.s_n4), R("s", N)))
        e("s_lshl_b32 %s, %s, 10" % (R("s", self.s_ldsA), R("s", self.s_w)), "this wave's 8-pixel block of every 32 pixels")
        # ---- A: base = in + first*TP*K*2 ; records = npix*K*2 - that
        tin = c.TP * c.K * 2
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", S_first), tin))
        e("s_mul_hi_u32 %s, %s, %d" % (R("s", t1), R("s", S_first), tin))
        e("s_mul_i32 %s, %s, %d" % (R("s", t2), R("s", M), c.K * 2))
        self.desc_from(self.srdA, ka + 0, t0, t1, t2, "A: this run's pixels")
        if c.bnin:
            self.desc_from(self.srdA2, ka + 8, t0, t1, t2, "a out: the same pixels of the other tensor")
            # its ReLU bits: 1 byte per 16 bytes
            e("s_lshr_b32 %s, %s, 4" % (R("s", t0), R("s", t0)))
            e("s_lshl_b32 %s, %s, 28" % (R("s", t3), R("s", t1)))
            e("s_or_b32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t3)))
            e("s_lshr_b32 %s, %s, 4" % (R("s", t1), R("s", t1)))
            e("s_lshr_b32 %s, %s, 4" % (R("s", t2), R("s", t2)))
            self.desc_from(self.srdBt, ka + 10, t0, t1, t2, "its ReLU bits")
            e("s_mov_b32 %s, %s" % (R("s", self.srdSS), R("s", ka + 12)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdSS + 1), R("s", ka + 13)))
            e("s_mov_b32 %s, %d" % (R("s", self.srdSS + 2), 2 * c.K * 4))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdSS + 3))
            e("s_mov_b32 %s, 0x00010001" % R("s", self.s_k1))
            # only the workgroups of column tile 0 store a and its bits (every column tile transforms its own LDS copy)
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_ct))
            e("s_cselect_b32 %s, 0, 0x80000000" % R("s", self.s_trflag))
        # ---- B: this wave's NT*16 weight rows
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_ct), c.BN * c.K * 2))
        e("s_mul_i32 %s, %s, %d" % (R("s", t1), R("s", self.s_wn), c.NT * 16 * c.K * 2))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdB), R("s", ka + 2), R("s", t0)))
        e("s_addc_u32 %s, %s, 0" % (R("s", self.srdB + 1), R("s", ka + 3)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdB + 1), R("s", self.srdB + 1)))
        e("s_mov_b32 %s, %d" % (R("s", self.srdB + 2), c.NT * 16 * c.K * 2))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdB + 3))
        # B lane offsets: tile n = 2p + odd, MFMA row rho = lane & 15 -> channel p*32 + (rho >> 2)*8 + odd*4 + (rho & 3); k bytes kg*16
        vb = [v[4], v[5]]
        e("v_lshrrev_b32 %s, 2, %s" % (R("v", v[6]), R("v", r)))
        e("v_and_b32 %s, 3, %s" % (R("v", v[7]), R("v", r)))
        e("v_lshl_add_u32 %s, %s, 3, %s" % (R("v", v[6]), R("v", v[6]), R("v", v[7])), "(rho >> 2)*8 + (rho & 3)")
        for odd in range(2):
            e("v_add_u32 %s, %d, %s" % (R("v", v[7]), 4 * odd, R("v", v[6])))
            e("v_mov_b32 %s, %d" % (R("v", v[8]), c.K * 2))
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", v[7]), R("v", v[7]), R("v", v[8])))
            e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", vb[odd]), R("v", kg), R("v", v[7])))
        self.comment("the weight slab: NT x KS fragments, once")
        for n in range(c.NT):
            p, odd = n >> 1, n & 1
            if odd == 0:
                e("s_mov_b32 %s, %d" % (R("s", t0), p * 32 * c.K * 2))
            for ks in range(c.KS):
                e("buffer_load_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("a", self.breg(n, ks), 4), R("v", vb[odd]), R("s", self.srdB, 4), R("s", t0), ks * 64))
        # ---- A DMA lane part: row-in-block = lane >> 3; logical chunk = (lane & 7) ^ ((row >> 1) & 7), row = (4j + w)*8 + (lane >> 3)
        l3, l7, j = v[6], v[7], v[8]
        e("v_lshrrev_b32 %s, 3, %s" % (R("v", l3), R("v", lane)))
        e("v_and_b32 %s, 7, %s" % (R("v", l7), R("v", lane)))
        e("v_lshrrev_b32 %s, 4, %s" % (R("v", j), R("v", lane)))
        e("s_and_b32 %s, %s, 1" % (R("s", t0), R("s", self.s_w)))
        e("s_lshl_b32 %s, %s, 2" % (R("s", t0), R("s", t0)), "+ 4 for the odd blocks (block = 4j + w)")
        e("v_or_b32 %s, %s, %s" % (R("v", j), R("s", t0), R("v", j)))
        e("v_xor_b32 %s, %s, %s" % (R("v", j), R("v", l7), R("v", j)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", j), R("v", j)))
        if c.bnin:
            e("v_lshlrev_b32 %s, 1, %s" % (R("v", self.v_ssoff), R("v", j)), "this lane's 16 bytes of a pixel's plane = 8 channels = 32 bytes of scale / shift")
            e("v_lshlrev_b32 %s, 4, %s" % (R("v", self.v_lane16), R("v", lane)))
        e("v_mov_b32 %s, %d" % (R("v", v[9]), c.K * 2))
        e("v_mad_u32_u24 %s, %s, %s, %s" % (R("v", self.vA_dma), R("v", l3), R("v", v[9]), R("v", j)))
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_w), 8 * c.K * 2))
        e("v_add_u32 %s, %s, %s" % (R("v", self.vA_dma), R("s", t0), R("v", self.vA_dma)))
        if c.bnin:
            for pl in range(c.NPL):
                for h in range(2):
                    e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", self.v_sc[pl] + 4 * h, 4), R("v", self.v_ssoff), R("s", self.srdSS, 4), pl * 256 + 16 * h))
                    e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", self.v_sh[pl] + 4 * h, 4), R("v", self.v_ssoff), R("s", self.srdSS, 4), c.K * 4 + pl * 256 + 16 * h))
        self.comment("A tile of the first tile")
        for jj in range(c.NPW):
            for ins in self.a_piece(jj, 0):
                e(ins)
        for ins in self.a_advance():
            e(ins)
        # ---- output-shaped tensors: base offset = first*tout + ct*BN*2 (64-bit), records = npix*N*2 - that
        e("s_mul_i32 %s, %s, %s" % (R("s", t0), R("s", S_first), R("s", self.s_tout)))
        e("s_mul_hi_u32 %s, %s, %s" % (R("s", t1), R("s", S_first), R("s", self.s_tout)))
        e("s_mul_i32 %s, %s, %d" % (R("s", t2), R("s", self.s_ct), c.BN * 2))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t2)))
        e("s_addc_u32 %s, %s, 0" % (R("s", t1), R("s", t1)))
        e("s_mul_i32 %s, %s, %s" % (R("s", t2), R("s", M), R("s", N)))
        e("s_lshl_b32 %s, %s, 1" % (R("s", t2), R("s", t2)), "bytes of an output-shaped tensor")
        self.desc_from(self.srdO, ka + 4, t0, t1, t2, "O")
        if c.stats >= 2:
            self.desc_from(self.srdY, ka + 8, t0, t1, t2, "y of the BN layer")
        if c.add == 3:
            # the whole half-resolution tensor from this column tile's first column on: npix / 4 pixels
            e("s_mul_i32 %s, %s, %d" % (R("s", t3), R("s", self.s_ct), c.BN * 2))
            e("s_add_u32 %s, %s, %s" % (R("s", self.srdAD), R("s", ka + 16), R("s", t3)))
            e("s_addc_u32 %s, %s, 0" % (R("s", self.srdAD + 1), R("s", ka + 17)))
            e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdAD + 1), R("s", self.srdAD + 1)))
            e("s_lshr_b32 %s, %s, 2" % (R("s", self.srdAD + 2), R("s", t2)))
            e("s_sub_u32 %s, %s, %s" % (R("s", self.srdAD + 2), R("s", self.srdAD + 2), R("s", t3)))
            e("s_mov_b32 %s, 0x00020000" % R("s", self.srdAD + 3))
            e("s_mul_i32 %s, %s, %d" % (R("s", self.s_qpf), R("s", S_first), c.TP))
        elif c.add:
            self.desc_from(self.srdAD, ka + 16, t0, t1, t2, "addend")
        if c.stats >= 2 or c.add == 2:
            # mask bytes: 1/16 of the byte offsets
            e("s_lshr_b32 %s, %s, 4" % (R("s", t0), R("s", t0)))
            e("s_lshl_b32 %s, %s, 28" % (R("s", t3), R("s", t1)))
            e("s_or_b32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t3)))
            e("s_lshr_b32 %s, %s, 4" % (R("s", t1), R("s", t1)))
            e("s_lshr_b32 %s, %s, 4" % (R("s", t2), R("s", t2)))
            if c.stats >= 2:
                self.desc_from(self.srdM, ka + 10, t0, t1, t2, "ReLU bits of the BN layer")
            if c.add == 2:
                self.desc_from(self.srdAB, ka + 18, t0, t1, t2, "ReLU bits of the addend")
        # ---- lane offsets of the output-shaped tensors: (m*16 + r)*N*2 + (w*NT*16 + kg*8)*2
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_wn), c.NT * 16 * 2))
        e("v_lshl_add_u32 %s, %s, 4, %s" % (R("v", v[6]), R("v", kg), R("s", t0)))
        e("s_lshl_b32 %s, %s, 1" % (R("s", t1), R("s", N)))
        rp, lanep = r, lane   # pixel of the tile: fragment row / lane, + this wave's first fragment
        if c.WM > 1:
            rp, lanep = v[10], v[11]
            e("s_mul_i32 %s, %s, %d" % (R("s", t3), R("s", self.s_wm), c.MFRW * 16))
            e("v_add_u32 %s, %s, %s" % (R("v", rp), R("s", t3), R("v", r)))
            e("v_add_u32 %s, %s, %s" % (R("v", lanep), R("s", t3), R("v", lane)))
        for m in range(c.MFRW):
            e("v_add_u32 %s, %d, %s" % (R("v", v[7]), 16 * m, R("v", rp)))
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", v[7]), R("v", v[7]), R("s", t1)))
            e("v_add_u32 %s, %s, %s" % (R("v", self.v_out_m[m]), R("v", v[7]), R("v", v[6])))
            if self.v_bits_m:
                e("v_lshrrev_b32 %s, 4, %s" % (R("v", self.v_bits_m[m]), R("v", self.v_out_m[m])), "mask bytes: one per 16-byte vector")
        e("v_lshlrev_b32 %s, 5, %s" % (R("v", self.v_chan), R("v", kg)), "this lane's 8 floats of a per-channel row")
        if c.NMT:
            e("s_lshr_b32 %s, %s, 3" % (R("s", t3), R("s", N)), "mask bytes of a pixel row")
            e("v_mul_lo_u32 %s, %s, %s" % (R("v", self.v_mk), R("v", lanep), R("s", t3)))
            e("s_mul_i32 %s, %s, %d" % (R("s", t3), R("s", self.s_wn), c.NT * 2))
            e("v_add_u32 %s, %s, %s" % (R("v", self.v_mk), R("s", t3), R("v", self.v_mk)), "lane = pixel of the tile: its mask bytes for this wave's columns")
            e("v_lshlrev_b32 %s, 2, %s" % (R("v", self.v_bp), R("v", r)))
            e("v_lshlrev_b32 %s, 3, %s" % (R("v", self.v_kg8), R("v", kg)))
        if c.add == 3:
            e("v_mov_b32 %s, %s" % (R("v", self.v_r), R("v", rp)))
            e("v_mov_b32 %s, %s" % (R("v", self.v_col), R("v", v[6])))
        if c.FULL:
            # full-line stores: after the lane exchange of the epilogue a lane holds, of pixel m*16 + (r & 7) [store 1] and of pixel
            # m*16 + 8 + (r & 7) [store 2], the 16 bytes at w*128 + (r >> 3)*64 + kg*16: 8 lanes cover a pixel's 128 bytes
            e("s_lshl_b32 %s, %s, 4" % (R("s", self.s_8rows), R("s", N)), "8 pixel rows of an output-shaped tensor")
            e("s_mov_b32 %s, 0x00ff00ff" % R("s", self.s_lo8))
            e("s_mov_b32 %s, 0x00ff00ff" % R("s", self.s_lo8 + 1))
            e("v_lshrrev_b32 %s, 3, %s" % (R("v", v[8]), R("v", r)))
            e("v_lshlrev_b32 %s, 6, %s" % (R("v", v[8]), R("v", v[8])))
            e("v_add_u32 %s, %s, %s" % (R("v", v[8]), R("v", v[8]), R("v", v[6])), "w*128 + kg*16 + (r >> 3)*64")
            e("v_and_b32 %s, 7, %s" % (R("v", v[9]), R("v", r)))
            for m in range(c.MFRW):
                e("v_add_u32 %s, %d, %s" % (R("v", v[7]), 16 * m, R("v", v[9])))
                e("v_mul_lo_u32 %s, %s, %s" % (R("v", v[7]), R("v", v[7]), R("s", t1)))
                e("v_add_u32 %s, %s, %s" % (R("v", self.v_st_m[m]), R("v", v[7]), R("v", v[8])))
        # ---- first tile's epilogue operands, then the prefetch descriptors move one tile ahead
        e("s_waitcnt lgkmcnt(0)")
        for m in range(c.MFRW):
            for ins in self.sub2_addr(m):
                e(ins)
        for ins in self.mask_loads(0):
            e(ins)
        for i in range(c.NI):
            for ins in self.item_loads(i):
                e(ins)
        for ins in self.prefetch_advance():
            e(ins)
        # ---- A read bases: row r of a fragment, chunk (kg + 4kk) ^ ((r >> 1) & 7)
        sw, cc = v[6], v[7]
        e("v_bfe_u32 %s, %s, 1, 3" % (R("v", sw), R("v", r)))
        e("v_xor_b32 %s, %s, %s" % (R("v", cc), R("v", kg), R("v", sw)))
        e("v_lshlrev_b32 %s, 4, %s" % (R("v", cc), R("v", cc)))
        e("v_lshl_add_u32 %s, %s, 7, %s" % (R("v", cc), R("v", r), R("v", cc)))
        if c.WM > 1:
            e("s_mul_i32 %s, %s, %d" % (R("s", t3), R("s", self.s_wm), c.MFRW * 2048))
            e("v_add_u32 %s, %s, %s" % (R("v", cc), R("s", t3), R("v", cc)), "this wave's first fragment")
        for b in range(c.NBUF):
            e("v_add_u32 %s, %d, %s" % (R("v", self.vA_rd[b][0]), b * c.ABUF, R("v", cc)))
            e("v_xor_b32 %s, 64, %s" % (R("v", self.vA_rd[b][1]), R("v", self.vA_rd[b][0])))
        if c.stats:
            for p in range(c.NT // 2):
                for i in range(8):
                    e("v_mov_b32 %s, 0" % R("v", self.s1[p] + i))
                    e("v_mov_b32 %s, 0" % R("v", self.s2[p] + i))
        e("s_waitcnt vmcnt(0)", "weights, first tile, first operands")

    # ---- Cfg.bnin: BatchNorm + ReLU of a landed tile in LDS ----------------------------------------------------------------------------------------------
    def tr_read(self, j, buf, i):
        c, r = self.c, self.tr[i]
        rb4 = c.TP // 8 // 4
        blk4, plane = j % rb4, j // rb4
        return ["s_add_u32 %s, %s, %d" % (R("s", self.s_ta), R("s", self.s_ldsA), buf * c.ABUF + plane * c.PLANE + blk4 * 4096),
                "v_add_u32 %s, %s, %s" % (R("v", r["ta"]), R("s", self.s_ta), R("v", self.v_lane16)),
                "ds_read_b128 %s, %s" % (R("v", r["d"], 4), R("v", r["ta"]))]

    def tr_work(self, j, i):
        """a = relu(y * scale + shift) of piece j (the 16 bytes this lane's LDS-DMA lane wrote) in register set i: back to LDS, to the `a` tensor at the
        LDS-DMA's own offset and one byte of ReLU bits at that offset / 16 (out of range, hence dropped, unless this workgroup stores: s_trflag)"""
        c, r = self.c, self.tr[i]
        rb4 = c.TP // 8 // 4
        blk4, plane = j % rb4, j // rb4
        d, f, t = r["d"], r["f"], r["f"] + 4
        sc, sh = self.v_sc[plane], self.v_sh[plane]
        g = ["v_add_u32 %s, %d, %s" % (R("v", r["o"]), blk4 * 32 * c.K * 2 + plane * 128, R("v", self.vA_dma)),
             "v_or_b32 %s, %s, %s" % (R("v", r["o"]), R("s", self.s_trflag), R("v", r["o"]))]
        for q in range(4):
            g += ["v_lshlrev_b32 %s, 16, %s" % (R("v", f + 2 * q), R("v", d + q)), "v_and_b32 %s, 0xffff0000, %s" % (R("v", f + 2 * q + 1), R("v", d + q))]
        for q in range(8):
            g.append("v_fma_f32 %s, %s, %s, %s" % (R("v", f + q), R("v", f + q), R("v", sc + q), R("v", sh + q)))
        for q in range(4):   # round first, ReLU on the packed pairs as signed 16-bit integers: the values of ReLU in fp32 followed by the rounding
            g.append("v_cvt_pk_bf16_f32 %s, %s, %s" % (R("v", f + q), R("v", f + 2 * q), R("v", f + 2 * q + 1)))
        for q in range(4):
            g.append("v_pk_max_i16 %s, %s, 0" % (R("v", f + q), R("v", f + q)))
        for q in range(4):   # ReLU bits = (a != 0): 0 / 1 per half, gathered to bit 2q + half of one byte
            g.append("v_pk_min_u16 %s, %s, %s" % (R("v", t + q), R("v", f + q), R("s", self.s_k1)))
        g += ["v_lshl_or_b32 %s, %s, 2, %s" % (R("v", t), R("v", t + 1), R("v", t)), "v_lshl_or_b32 %s, %s, 2, %s" % (R("v", t + 2), R("v", t + 3), R("v", t + 2)),
              "v_lshl_or_b32 %s, %s, 4, %s" % (R("v", t), R("v", t + 2), R("v", t)), "v_lshrrev_b32 %s, 4, %s" % (R("v", r["o2"]), R("v", r["o"])),
              "v_lshrrev_b32 %s, 15, %s" % (R("v", t + 1), R("v", t)), "v_and_b32 %s, 0x55, %s" % (R("v", t), R("v", t)),
              "v_and_b32 %s, 0xaa, %s" % (R("v", t + 1), R("v", t + 1)), "v_or_b32 %s, %s, %s" % (R("v", r["bits"]), R("v", t + 1), R("v", t)),
              "ds_write_b128 %s, %s" % (R("v", r["ta"]), R("v", f, 4)),
              "buffer_store_dwordx4 %s, %s, %s, 0 offen" % (R("v", f, 4), R("v", r["o"]), R("s", self.srdA2, 4)),
              "buffer_store_byte %s, %s, %s, 0 offen" % (R("v", r["bits"]), R("v", r["o2"]), R("s", self.srdBt, 4))]
        return g

    def transform_tile(self, buf, first):
        """this wave's NPW pieces of the tile in A buffer `buf` (they have landed: the caller waited for this wave's own LDS-DMA), the next piece's
        read under the work on this one; then the a / bits descriptors move on to the next tile while one is left (else the stores are switched
        off: the look-ahead tile behind the run's last one is a re-read that nobody uses)"""
        c, e = self.c, self.e
        if not first:
            # a real tile?  (the stores of the workgroups of column tiles > 0 are off for good)
            e("s_cmp_gt_u32 %s, 0" % R("s", self.s_trleft))
            e("s_cselect_b32 %s, %s, 0x80000000" % (R("s", self.s_trflag), R("s", self.s_trflag)))
        for ins in self.tr_read(0, buf, 0):
            e(ins)
        for j in range(c.NPW):
            e("s_waitcnt lgkmcnt(%d)" % (1 if j else 0))
            if j + 1 < c.NPW:
                for ins in self.tr_read(j + 1, buf, (j + 1) & 1):
                    e(ins)
            for ins in self.tr_work(j, j & 1):
                e(ins)
        # descriptors one tile on (prologue: if the run has a second tile; loop: if another real tile follows this one)
        e("s_cmp_gt_u32 %s, %d" % (R("s", self.s_trleft), 0 if first else 1))
        e("s_cselect_b32 %s, %d, 0" % (R("s", self.s_tinc), c.TP * c.K * 2))
        e("s_cselect_b32 %s, %d, 0" % (R("s", self.s_tincb), c.TP * c.K * 2 // 16))
        for ins in self.desc_adv(self.srdA2, self.s_tinc) + self.desc_adv(self.srdBt, self.s_tincb):
            e(ins)
        if not first:
            e("s_cmp_gt_u32 %s, 0" % R("s", self.s_trleft))
            e("s_cselect_b32 %s, 1, 0" % R("s", self.s_t3))
            e("s_sub_u32 %s, %s, %s" % (R("s", self.s_trleft), R("s", self.s_trleft), R("s", self.s_t3)))
        e("s_waitcnt lgkmcnt(0)")

    def a_advance(self):
        """the A descriptor moves to the next tile while one is left, else stays (re-reads of the last tile, never used)"""
        c = self.c
        return ["s_cmp_gt_u32 %s, 0" % R("s", self.s_pfa),
                "s_cselect_b32 %s, %d, 0" % (R("s", self.s_ia), c.TP * c.K * 2),
                "s_cselect_b32 %s, 1, 0" % R("s", self.s_t3),
                "s_sub_u32 %s, %s, %s" % (R("s", self.s_pfa), R("s", self.s_pfa), R("s", self.s_t3))] + self.desc_adv(self.srdA, self.s_ia)

    def prefetch_advance(self):
        """the prefetch descriptors of the epilogue operands (y, addend, masks) move to the next tile while one is left, else stay"""
        c = self.c
        if not c.L:
            return []
        out = ["s_cmp_gt_u32 %s, 0" % R("s", self.s_pf),
               "s_cselect_b32 %s, %s, 0" % (R("s", self.s_io), R("s", self.s_tout)),
               "s_cselect_b32 %s, %s, 0" % (R("s", self.s_ib), R("s", self.s_tbits)),
               "s_cselect_b32 %s, 1, 0" % R("s", self.s_t3),
               "s_sub_u32 %s, %s, %s" % (R("s", self.s_pf), R("s", self.s_pf), R("s", self.s_t3))]
        if c.add == 3:
            out += ["s_mul_i32 %s, %s, %d" % (R("s", self.s_t3), R("s", self.s_t3), c.TP),
                    "s_add_u32 %s, %s, %s" % (R("s", self.s_qpf), R("s", self.s_qpf), R("s", self.s_t3))]
        if c.stats >= 2:
            out += self.desc_adv(self.srdY, self.s_io) + self.desc_adv(self.srdM, self.s_ib)
        if c.add in (1, 2):
            out += self.desc_adv(self.srdAD, self.s_io)
        if c.add == 2:
            out += self.desc_adv(self.srdAB, self.s_ib)
        return out

    # -----------------------------------------------------------------------------------------------------------------
    @staticmethod
    def spread(mf, others, first=1):
        """MFMA list with the instruction groups `others` placed evenly between them"""
        n, k = len(mf), len(others)
        slots = {}
        if k:
            span = max(1, n - first - 1)
            for j, grp in enumerate(others):
                slots.setdefault(min(n - 1, first + (j * span) // k), []).extend(grp)
        out = []
        for i, m in enumerate(mf):
            out.append(m)
            out.extend(slots.get(i, []))
        return out

    def capture(self, fn, *args):
        """the instructions fn(*args) emits, as a list"""
        keep, self.out = self.out, []
        fn(*args)
        got, self.out = self.out, keep
        return got

    def mfma_phase(self, buf, aset):
        """tile in A buffer buf -> accumulator set aset: KS k-steps of MFR x NT MFMAs, the fragment reads of step ks + 1 between the MFMAs of step ks"""
        c, e = self.c, self.e
        for ins in self.frag_reads(0, 0, buf):
            e(ins)
        for ks in range(c.KS):
            e("s_waitcnt lgkmcnt(0)")
            fs = self.F[ks & 1]
            mf = []
            for n in range(c.NT):
                for m in range(c.MFRW):
                    a = self.acc(m, n, aset)
                    csrc = "0" if ks == 0 else R("a", a, 4)
                    mf.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (R("a", a, 4), R("a", self.breg(n, ks), 4), R("v", fs + 4 * m, 4), csrc))
            if c.probe & 1:
                mf = mf[:c.MFRW] if ks == 0 else []
            nxt = [[r] for r in self.frag_reads((ks + 1) & 1, ks + 1, buf)] if ks + 1 < c.KS else []
            for ins in (self.spread(mf, nxt) if mf else [x for g in nxt for x in g]):
                e(ins)

    def epilogue_tile(self, CW):
        c = self.c
        for ins in self.mask_loads(self.mbuf ^ 1):   # the NEXT tile's mask bytes (waited for by the next tile's first counted wait: they are older)
            self.e(ins)
        if c.FULL:
            for m in range(c.MFRW):
                self.unit(m, CW)
        else:
            for i in range(c.NI):
                m, p = divmod(i, c.NT // 2)
                self.item(i, m, p, CW)

    @staticmethod
    def weave(primary, secondary):
        """secondary (the MFMA phase of the next tile, order kept) spread evenly through primary (the epilogue of this tile, order kept).  An
        s_waitcnt of either list keeps its place relative to its own list: lgkmcnt waits belong to the MFMA phase's fragment reads (the
        epilogue has no LDS operation), vmcnt waits to the epilogue's loads (the MFMA phase has no vector-memory operation)."""
        np_, ns = len(primary), len(secondary)
        out, j = [], 0
        for i, ins in enumerate(primary):
            out.append(ins)
            while j < ns and (j + 1) * np_ <= (i + 1) * ns:
                out.append(secondary[j])
                j += 1
        out.extend(secondary[j:])
        return out

    def mainloop(self):
        c, e = self.c, self.e
        top, done = self.newlabel("loop"), self.newlabel("done")
        # younger than an item's (a fragment's) loads when it waits for them: the stores + refills of the other items (fragments) of a
        # tile and the A pieces requested at the top of this tile's trip
        CW = ((c.L + 1) * (c.NI - 1) if not c.FULL else 2 * (c.L + 1) * (c.MFRW - 1)) + c.NPW + c.NMT
        TOPW = c.NI * (c.L + 1) + c.NMT   # younger than a tile's A pieces at the top of the trip that needs them
        if c.bnin:
            TOPW += 2 * c.NPW             # ... and the a / bits stores of the transform, which runs BEHIND the request of the tile after next
        if c.weave:
            self.comment("---- first tile: its MFMAs alone (A buffer 0 -> accumulator set 0), the second tile requested")
            if c.bnin:
                e("s_sub_u32 %s, %s, 1" % (R("s", self.s_trleft), R("s", self.s_cnt)), "tiles behind the first")
                self.transform_tile(0, True)
            e("s_barrier")
            for jj in range(c.NPW):
                for ins in self.a_piece(jj, 1):
                    e(ins)
            for ins in self.a_advance():
                e(ins)
            self.mfma_phase(0, 0)
            e("s_waitcnt vmcnt(0)", "the second tile has landed (the counted wait at the top of the first trip has nothing younger to count yet)")
            self.comment("---- main loop: per trip the epilogue of tile t out of one accumulator set, the MFMAs of tile t + 1 into the other woven through it")
        else:
            self.comment("---- main loop: one 64-pixel tile per trip (unrolled over the %d A buffers)" % c.NBUF)
        self.label(top)
        for b in range(c.NBUF):
            nb = (b + 1) % c.NBUF
            self.comment("tile in A buffer %d" % b)
            if c.weave:
                # at the top of the trip of tile t: tile t + 1 (requested a trip ago) has landed for every wave; buffer b (tile t: its MFMAs ran
                # in the previous trip) takes tile t + 2
                e("s_waitcnt vmcnt(%d)" % TOPW, "tile t + 1's A pieces have landed (younger: the last trip's stores and refills)")
                e("s_barrier")
                for jj in range(c.NPW):
                    for ins in self.a_piece(jj, b):
                        e(ins)
                for ins in self.a_advance():
                    e(ins)
                if c.bnin:
                    # tile t + 2 is on its way: the transform of tile t + 1 runs under that latency (in front of the request it lengthened the
                    # chain wait -> request by 300 cycles per piece: +15 us per launch); a second barrier publishes it
                    self.transform_tile(nb, False)
                    e("s_barrier")
                self.accset = b
                self.mbuf = b
                epi = self.capture(self.epilogue_tile, CW)
                mfm = self.capture(self.mfma_phase, nb, nb)
                for ins in self.weave(epi, mfm):
                    self.out.append(ins)
            else:
                e("s_waitcnt vmcnt(%d)" % TOPW, "this tile's A pieces have landed (younger: the last tile's stores and refills)")
                e("s_barrier")
                for jj in range(c.NPW):
                    for ins in self.a_piece(jj, nb):
                        e(ins)
                for ins in self.a_advance():
                    e(ins)
                self.accset = 0
                self.mbuf = b
                self.mfma_phase(b, 0)
                # (the accumulators are read by VALU instructions: the matrix pipe must have retired the last MFMAs)
                e("s_nop 15")
                e("s_nop 7")
                self.epilogue_tile(CW)
            # ---- next tile
            for ins in self.desc_adv(self.srdO, self.s_tout):
                e(ins)
            for ins in self.prefetch_advance():
                e(ins)
            e("s_sub_u32 %s, %s, 1" % (R("s", self.s_cnt), R("s", self.s_cnt)))
            e("s_cmp_eq_u32 %s, 0" % R("s", self.s_cnt))
            if b < c.NBUF - 1:
                e("s_cbranch_scc1 %s" % done)
            else:
                e("s_cbranch_scc0 %s" % top)
        self.label(done)

    def pair_math(self, i, m, p, ds):
        """accumulators of (fragment m, tile pair p) -> + addend -> bf16 in ds; statistics of the pair"""
        c, e = self.c, self.e
        d = self.it[i]
        tv, xr, yv, vm = self.tv, self.xr, self.yv, self.v_m
        for k in range(4):
            e("v_accvgpr_read_b32 %s, a%d" % (R("v", tv + k), self.acc(m, 2 * p) + k))
            e("v_accvgpr_read_b32 %s, a%d" % (R("v", tv + 4 + k), self.acc(m, 2 * p + 1) + k))
        if c.add and not (c.probe & 2):
            for k in range(8):
                src = R("v", d["ad"] + k // 2)
                if k & 1:
                    e("v_and_b32 %s, 0xffff0000, %s" % (R("v", xr + k), src))
                else:
                    e("v_lshlrev_b32 %s, 16, %s" % (R("v", xr + k), src))
                if c.add == 2:
                    e("v_bfe_i32 %s, %s, %d, 1" % (R("v", vm), R("v", d["ab"]), k), "0 / -1: ReLU bit of addend element %d" % k)
                    e("v_and_b32 %s, %s, %s" % (R("v", xr + k), R("v", xr + k), R("v", vm)))
                e("v_add_f32 %s, %s, %s" % (R("v", tv + k), R("v", tv + k), R("v", xr + k)))
        for k in range(4):
            e("v_cvt_pk_bf16_f32 %s, %s, %s" % (R("v", ds + k), R("v", tv + 2 * k), R("v", tv + 2 * k + 1)))
        if c.stats and not (c.probe & 2):
            s1, s2 = self.s1[p], self.s2[p]
            for k in range(4):
                e("v_lshlrev_b32 %s, 16, %s" % (R("v", xr + 2 * k), R("v", ds + k)))
                e("v_and_b32 %s, 0xffff0000, %s" % (R("v", xr + 2 * k + 1), R("v", ds + k)))
            if c.stats == 1:
                for k in range(8):
                    e("v_add_f32 %s, %s, %s" % (R("v", s1 + k), R("v", s1 + k), R("v", xr + k)))
                    e("v_fma_f32 %s, %s, %s, %s" % (R("v", s2 + k), R("v", xr + k), R("v", xr + k), R("v", s2 + k)))
            else:
                for k in range(8):
                    e("v_bfe_i32 %s, %s, %d, 1" % (R("v", vm), R("v", d["yb"]), k), "0 / -1: ReLU bit of element %d" % k)
                    if c.stats == 3:   # leaky mask: dz = bit ? dx : dx * 0.01 (dconv_gen.py LEAKY_BITS; tv + k is dead after the conversion)
                        e("v_mul_f32 %s, 0x%08x, %s" % (R("v", tv + k), LEAKY_BITS, R("v", xr + k)))
                        e("v_bfi_b32 %s, %s, %s, %s" % (R("v", xr + k), R("v", vm), R("v", xr + k), R("v", tv + k)), "dz")
                    else:
                        e("v_and_b32 %s, %s, %s" % (R("v", xr + k), R("v", xr + k), R("v", vm)), "dz")
                    if k & 1:
                        e("v_and_b32 %s, 0xffff0000, %s" % (R("v", yv + 1), R("v", d["y"] + k // 2)))
                    else:
                        e("v_lshlrev_b32 %s, 16, %s" % (R("v", yv), R("v", d["y"] + k // 2)))
                    e("v_add_f32 %s, %s, %s" % (R("v", s1 + k), R("v", s1 + k), R("v", xr + k)))
                    e("v_fma_f32 %s, %s, %s, %s" % (R("v", s2 + k), R("v", xr + k), R("v", yv + (k & 1)), R("v", s2 + k)), "sum dz*y")

    def unit(self, m, CW):
        """fragment m, both tile pairs: a lane ends with the two 16-byte chunks (pair 0, pair 1) of pixel r; lanes r and r + 8 of a row
        swap one chunk each, so that store 1 writes pixels 0 .. 7 and store 2 pixels 8 .. 15 of the fragment as whole 128-byte lines
        (tools/micro/seg_read.hip: stores of 64-byte half lines run at 0.7 of the rate of whole lines)"""
        c, e = self.c, self.e
        i0, i1 = 2 * m, 2 * m + 1
        A, B, C = self.dsets[(2 * m) % 4], self.dsets[(2 * m + 1) % 4], self.v_xc
        self.comment("fragment %d" % m)
        if (c.L or c.NMT) and not (c.probe & 4):
            e("s_waitcnt vmcnt(%d)" % CW)
        for ins in self.mask_fetch(m, self.mbuf):
            e(ins)
        self.pair_math(i0, m, 0, A)
        self.pair_math(i1, m, 1, B)
        e("s_nop 1")
        for k in range(4):   # C = the partner lane's pair-0 chunk
            e("v_mov_b32_dpp %s, %s row_ror:8 row_mask:0xf bank_mask:0xf" % (R("v", C + k), R("v", A + k)))
        for k in range(4):   # lanes 8 .. 15: the partner's pair-1 chunk (pixel r - 8) -> store 1 holds pixels 0 .. 7
            e("v_mov_b32_dpp %s, %s row_ror:8 row_mask:0xf bank_mask:0xc" % (R("v", A + k), R("v", B + k)))
        for k in range(4):   # lanes 0 .. 7: the partner's pair-0 chunk (pixel r + 8) -> store 2 holds pixels 8 .. 15
            e("v_cndmask_b32 %s, %s, %s, %s" % (R("v", B + k), R("v", B + k), R("v", C + k), R("s", self.s_lo8, 2)))
        if not (c.probe & 8):
            vt = self.v_tmp[0]
            e("v_add_u32 %s, %s, %s" % (R("v", vt), R("s", self.s_8rows), R("v", self.v_st_m[m])))
            nt = " nt" if c.nt & 2 else ""
            e("buffer_store_dwordx4 %s, %s, %s, 0 offen%s" % (R("v", A, 4), R("v", self.v_st_m[m]), R("s", self.srdO, 4), nt))
            e("buffer_store_dwordx4 %s, %s, %s, 0 offen%s" % (R("v", B, 4), R("v", vt), R("s", self.srdO, 4), nt))
        for ins in self.sub2_addr(m) + self.item_loads(i0) + self.item_loads(i1):   # rolling refill: the same vectors of the NEXT tile
            e(ins)

    def item(self, i, m, p, CW):
        c, e = self.c, self.e
        d = self.it[i]
        tv, xr, yv, vm = self.tv, self.xr, self.yv, self.v_m
        ds = self.dsets[i % 4]
        self.comment("item %d: fragment %d, tile pair %d" % (i, m, p))
        if (c.L or c.NMT) and not (c.probe & 4):
            e("s_waitcnt vmcnt(%d)" % CW)
        if p == 0:
            for ins in self.mask_fetch(m, self.mbuf):
                e(ins)
        for k in range(4):
            e("v_accvgpr_read_b32 %s, a%d" % (R("v", tv + k), self.acc(m, 2 * p) + k))
            e("v_accvgpr_read_b32 %s, a%d" % (R("v", tv + 4 + k), self.acc(m, 2 * p + 1) + k))
        if c.add and not (c.probe & 2):
            for k in range(8):
                src = R("v", d["ad"] + k // 2)
                if k & 1:
                    e("v_and_b32 %s, 0xffff0000, %s" % (R("v", xr + k), src))
                else:
                    e("v_lshlrev_b32 %s, 16, %s" % (R("v", xr + k), src))
                if c.add == 2:
                    e("v_bfe_i32 %s, %s, %d, 1" % (R("v", vm), R("v", d["ab"]), k), "0 / -1: ReLU bit of addend element %d" % k)
                    e("v_and_b32 %s, %s, %s" % (R("v", xr + k), R("v", xr + k), R("v", vm)))
                e("v_add_f32 %s, %s, %s" % (R("v", tv + k), R("v", tv + k), R("v", xr + k)))
        for k in range(4):
            e("v_cvt_pk_bf16_f32 %s, %s, %s" % (R("v", ds + k), R("v", tv + 2 * k), R("v", tv + 2 * k + 1)))
        if not (c.probe & 8):
            e("buffer_store_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", ds, 4), R("v", self.v_out_m[m]), R("s", self.srdO, 4), p * 64))
        if c.stats and not (c.probe & 2):
            s1, s2 = self.s1[p], self.s2[p]
            for k in range(4):
                e("v_lshlrev_b32 %s, 16, %s" % (R("v", xr + 2 * k), R("v", ds + k)))
                e("v_and_b32 %s, 0xffff0000, %s" % (R("v", xr + 2 * k + 1), R("v", ds + k)))
            if c.stats == 1:
                for k in range(8):
                    e("v_add_f32 %s, %s, %s" % (R("v", s1 + k), R("v", s1 + k), R("v", xr + k)))
                    e("v_fma_f32 %s, %s, %s, %s" % (R("v", s2 + k), R("v", xr + k), R("v", xr + k), R("v", s2 + k)))
            else:
                for k in range(8):
                    e("v_bfe_i32 %s, %s, %d, 1" % (R("v", vm), R("v", d["yb"]), k), "0 / -1: ReLU bit of element %d" % k)
                    if c.stats == 3:   # leaky mask: dz = bit ? dx : dx * 0.01 (dconv_gen.py LEAKY_BITS; tv + k is dead after the conversion)
                        e("v_mul_f32 %s, 0x%08x, %s" % (R("v", tv + k), LEAKY_BITS, R("v", xr + k)))
                        e("v_bfi_b32 %s, %s, %s, %s" % (R("v", xr + k), R("v", vm), R("v", xr + k), R("v", tv + k)), "dz")
                    else:
                        e("v_and_b32 %s, %s, %s" % (R("v", xr + k), R("v", xr + k), R("v", vm)), "dz")
                    if k & 1:
                        e("v_and_b32 %s, 0xffff0000, %s" % (R("v", yv + 1), R("v", d["y"] + k // 2)))
                    else:
                        e("v_lshlrev_b32 %s, 16, %s" % (R("v", yv), R("v", d["y"] + k // 2)))
                    e("v_add_f32 %s, %s, %s" % (R("v", s1 + k), R("v", s1 + k), R("v", xr + k)))
                    e("v_fma_f32 %s, %s, %s, %s" % (R("v", s2 + k), R("v", xr + k), R("v", yv + (k & 1)), R("v", s2 + k)), "sum dz*y")
        for ins in (self.sub2_addr(m) if p == 0 else []) + self.item_loads(i):   # rolling refill: the same vectors of the NEXT tile
            e(ins)

    # -----------------------------------------------------------------------------------------------------------------
    def finale(self):
        c, e = self.c, self.e
        ka = self.s_ka
        t0, t1 = self.s_t0, self.s_t1
        npair = c.NT // 2
        self.comment("---- end of the run: statistics row of this workgroup")
        e("s_waitcnt vmcnt(0)")
        if not c.stats:
            e("s_endpgm")
            return
        # row g: [2][N] floats; this wave's channels start at ct*BN + w*NT*16
        e("s_mul_i32 %s, %s, %d" % (R("s", t0), R("s", self.s_ct), c.BN * 4))
        e("s_mul_i32 %s, %s, %d" % (R("s", t1), R("s", self.s_wn), c.NT * 16 * 4))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t1)), "byte offset of this wave's channels in a per-channel float row")
        mu = isd = None
        if c.stats >= 2:
            # mean / invstd of this lane's 8 channels per pair (the fragment registers are free)
            for name, k0 in (("mu", 12), ("is", 14)):
                srd = self.srdY if name == "mu" else self.srdM
                e("s_add_u32 %s, %s, %s" % (R("s", srd), R("s", ka + k0), R("s", t0)))
                e("s_addc_u32 %s, %s, 0" % (R("s", srd + 1), R("s", ka + k0 + 1)))
                e("s_and_b32 %s, %s, 0xffff" % (R("s", srd + 1), R("s", srd + 1)))
                e("s_mov_b32 %s, %d" % (R("s", srd + 2), c.NT * 16 * 4))
                e("s_mov_b32 %s, 0x00020000" % R("s", srd + 3))
            # registers: the (now dead) y vectors of the first items hold mean, the packed-output sets hold invstd, 4 floats per block
            mu = [[self.it[2 * p + h]["y"] + k for h in range(2) for k in range(4)] for p in range(npair)]
            isd = [[self.dsets[2 * p + h] + k for h in range(2) for k in range(4)] for p in range(npair)]
            assert 2 * npair <= c.NI and 2 * npair <= 4
            for p in range(npair):
                for h in range(2):
                    e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", mu[p][4 * h], 4), R("v", self.v_chan), R("s", self.srdY, 4), p * 128 + 16 * h))
                    e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", isd[p][4 * h], 4), R("v", self.v_chan), R("s", self.srdM, 4), p * 128 + 16 * h))
        # statistics row descriptor: stat + g*2*N*4 + channel offset
        if c.WM > 1:   # row g * WM + wm: the waves that share columns leave a row each
            e("s_lshl_b32 %s, %s, %d" % (R("s", t1), R("s", self.s_g), c.WM.bit_length() - 1))
            e("s_add_u32 %s, %s, %s" % (R("s", t1), R("s", t1), R("s", self.s_wm)))
            e("s_mul_i32 %s, %s, %s" % (R("s", t1), R("s", t1), R("s", self.s_n4)))
        else:
            e("s_mul_i32 %s, %s, %s" % (R("s", t1), R("s", self.s_g), R("s", self.s_n4)))
        e("s_lshl_b32 %s, %s, 1" % (R("s", t1), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", t0), R("s", t0), R("s", t1)))
        e("s_add_u32 %s, %s, %s" % (R("s", self.srdX), R("s", ka + 6), R("s", t0)))
        e("s_addc_u32 %s, %s, 0" % (R("s", self.srdX + 1), R("s", ka + 7)))
        e("s_and_b32 %s, %s, 0xffff" % (R("s", self.srdX + 1), R("s", self.srdX + 1)))
        e("s_mov_b32 %s, 0x7fffffff" % R("s", self.srdX + 2))
        e("s_mov_b32 %s, 0x00020000" % R("s", self.srdX + 3))
        for sh in (1, 2, 4, 8):
            for p in range(npair):
                for arr in (self.s1[p], self.s2[p]):
                    for k in range(8):
                        rr = R("v", arr + k)
                        e("v_add_f32_dpp %s, %s, %s row_shr:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (rr, rr, rr, sh))
        if c.stats >= 2:
            e("s_waitcnt vmcnt(0)")
            tv = self.tv
            for p in range(npair):
                # sum dz*xhat = invstd * (sum dz*y - mean * sum dz)
                for k in range(8):
                    e("v_mul_f32 %s, %s, %s" % (R("v", tv + k), R("v", mu[p][k]), R("v", self.s1[p] + k)))
                for k in range(8):
                    e("v_sub_f32 %s, %s, %s" % (R("v", self.s2[p] + k), R("v", self.s2[p] + k), R("v", tv + k)))
                for k in range(8):
                    e("v_mul_f32 %s, %s, %s" % (R("v", self.s2[p] + k), R("v", isd[p][k]), R("v", self.s2[p] + k)))
        e("s_mov_b32 exec_lo, 0x80008000", "lanes 15 of every row hold the row sums")
        e("s_mov_b32 exec_hi, 0x80008000")
        for p in range(npair):
            for h in range(2):
                e("buffer_store_dwordx4 %s, %s, %s, 0 offen offset:%d" % (R("v", self.s1[p] + 4 * h, 4), R("v", self.v_chan), R("s", self.srdX, 4), p * 128 + 16 * h))
                e("buffer_store_dwordx4 %s, %s, %s, %s offen offset:%d" % (R("v", self.s2[p] + 4 * h, 4), R("v", self.v_chan), R("s", self.srdX, 4), R("s", self.s_n4), p * 128 + 16 * h))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")

    # -----------------------------------------------------------------------------------------------------------------
    def finish(self):
        c = self.c
        name = c.name
        lds = c.LDS
        total_v = self.accum_offset + self.nagpr
        hdr = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6", "\t.text", "\t.protected\t%s" % name, "\t.globl\t%s" % name,
               "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
        tail = ["\t.section\t.rodata,\"a\",@progbits", "\t.p2align\t6, 0x0", "\t.amdhsa_kernel %s" % name]
        kd = dict(group_segment_fixed_size=lds, private_segment_fixed_size=0, kernarg_size=self.KA["size"],
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
        tail += ["\t.end_amdhsa_kernel", "\t.text", "\t.amdgpu_metadata", "---", "amdhsa.kernels:", "  - .agpr_count:     %d" % self.nagpr, "    .args:"]
        off = 0
        for i in range(10):
            tail.append("      - .address_space:  global\n        .offset:         %d\n        .size:           8\n        .value_kind:     global_buffer" % off)
            off += 8
        for i in range(6):
            tail.append("      - .offset:         %d\n        .size:           4\n        .value_kind:     by_value" % off)
            off += 4
        tail.append("      - .offset:         %d\n        .size:           %d\n        .value_kind:     by_value" % (off, self.KA["size"] - off))
        tail += ["    .group_segment_fixed_size: %d" % lds, "    .kernarg_segment_align: 8", "    .kernarg_segment_size: %d" % self.KA["size"],
                 "    .max_flat_workgroup_size: 256", "    .name:           %s" % name, "    .private_segment_fixed_size: 0",
                 "    .sgpr_count:     %d" % (self.S.n + 6), "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % name,
                 "    .uniform_work_group_size: 1", "    .uses_dynamic_stack: false", "    .vgpr_count:     %d" % total_v, "    .vgpr_spill_count: 0",
                 "    .wavefront_size: 64", "amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:\n  - 1\n  - 2", "...", "\t.end_amdgpu_metadata"]
        body = self.out + ["\t.p2align 8", ".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name)]
        self.lds_bytes = lds
        return "\n".join(hdr + body + tail) + "\n"


def _variants():
    v = {}
    for (K, BN) in ((64, 256), (128, 256), (256, 256), (512, 128), (256, 128), (256, 64), (64, 64)):
        for (st, add) in ((0, 0), (1, 0), (2, 0), (2, 1), (2, 2), (0, 1), (0, 2), (2, 3), (0, 3)):
            if (K, BN) == (256, 128) and add != 0:   # 256 -> 128 columns: layer 2's first conv1 forward (129 us against 159)
                continue
            if BN == 64 and add == 3:                # the 64-column launches of layer 1 (waves 2 x 2, 128-pixel tiles) meet no striding block
                continue
            name = "po_k%d_b%d_s%d_a%d" % (K, BN, st, add)
            v[name] = PoCfg(name, K=K, BN=BN, stats=st, add=add, **(dict(WM=2, MFR=8) if BN == 64 else {}))
    for (K, BN) in ((256, 64), (512, 128)):   # BASELINE configs[3] (BResNet-50): conv3's data gradient of layers 1 / 2 with bn2's backward sums under the leaky mask
        name = "po_k%d_b%d_s3_a0" % (K, BN)
        v[name] = PoCfg(name, K=K, BN=BN, stats=3, add=0, **(dict(WM=2, MFR=8) if BN == 64 else {}))
    for K in (64, 128, 256):   # conv3's forward of layers 1 - 3 with bn2 + ReLU in its operand path
        name = "po_k%d_b256_s1_a0_bn" % K
        v[name] = PoCfg(name, K=K, BN=256, stats=1, add=0, bnin=1)
    return v


VARIANTS = _variants()


def generate(base, **over):
    c = VARIANTS[base]
    if over:
        c = PoCfg(**{**c.__dict__, **over})
    g = Gen(c)
    return c, g, g.gen()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--set", action="append", default=[], help="tuning: override a PoCfg field (key=int), with --suffix names the kernel")
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
        with open(os.path.join(a.out, name + ".s"), "w") as f:
            f.write(text)
        print("%s: %d vgpr + %d agpr, %d sgpr, lds %d, %d lines" % (name, g.accum_offset, g.nagpr, g.S.n, g.lds_bytes, text.count("\n")))


if __name__ == "__main__":
    main()
