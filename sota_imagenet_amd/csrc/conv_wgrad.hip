// conv_wgrad.hip — convolution weight-gradient for gfx950 (MI355X): split-K MFMA GEMM + deterministic reduce.
//
// Replaces the cuDNN wgrad kernels under `loss.backward()` in the reference (sota_imagenet/callbacks.py:317).
//
//   dW[co][tap][ci] = sum over output pixels m of dy[m][co] * x[src(m,tap)][ci]
// is a GEMM whose reduction index (the pixel) is the SLOW index of both operands in NHWC memory, so both MFMA
// operands are "transposed".  Slabs of BKP pixels are staged in their natural [pixel][channel] layout by direct-to-LDS
// buffer loads (buffer_load_dwordx4 ... offen lds; per-lane byte offset = the tap's gathered pixel, out-of-image rows
// get an out-of-range offset and read zeros) and read TRANSPOSED out of LDS:
//   fp32 : ds_read_b32 (a lane holds ONE k per v_mfma_f32_32x32x2_f32 operand, lanes run along channels)
//   bf16 : ds_read_b64_tr_b16 x2 per v_mfma_f32_32x32x16_bf16 operand; the LDS image stays lane-linear and the four
//          pixel rows a half-wave touches are spread over all 64 banks by XOR-ing the 16-byte chunk index with
//          (row&3)<<2 (256-byte rows) / ((row>>1)&1)<<2 (128-byte rows) on the SOURCE address and on the read.
//   fp8  : (e4m3 twins of dy and x, fp8 training step) ONE ds_read_b64_tr_b8 per v_mfma_f32_32x32x16_fp8_fp8 operand: per
//          16-lane group the instruction reads a block of 8 rows x 16 byte-columns, lane 2q + p supplies the address of row q,
//          columns 8p .. 8p+7, lane i receives column i of the 8 rows (tools/micro/tr8_test.hip pins that on the device); the
//          8 rows x 32 bytes a half-wave touches cover the 64 banks once with the chunk index XORed by ((row>>1)&3)<<1.
// Tile: BMC (128|64) output channels x BNC (128|64) input channels of TPI taps (1; 3 or 4 for the 64x64 tiles of the
// small-channel layers, where one staged dy slab then feeds TPI x slabs: the 64x64 single-tap tile is fetch-bound),
// 4 waves (2x2), swapped MFMA operands
// so a lane ends with 4 consecutive input channels of one output channel (16-byte stores).  Persistent workgroups
// walk (tile, pixel-split, slab) as one stream through a 2-stage LDS ring with the next slab always in flight
// (counted s_waitcnt vmcnt + raw s_barrier, as in conv_igemm.hip).  fp32 partial slabs are summed in split order by
// splitk_reduce (bitwise reproducible, no float atomics).
#include <cstdlib>

#include "common.h"
#include "lds_dma.h"

namespace mi355 {

namespace {

#define MAX_WG (2 * device_cus())  // persistent workgroups: 2 per CU
#ifndef MI355_WGRAD_BKP16
#define MI355_WGRAD_BKP16 64  // pixels per bf16 slab (A/B builds: -DMI355_WGRAD_BKP16=32 halves the LDS per workgroup)
#endif

struct FastDiv {
  uint32_t mul, sh;
};
static FastDiv make_fastdiv(uint32_t d) {  // exact for 0 <= n < 2^31
  uint32_t L = 0;
  while ((1u << L) < d) ++L;
  FastDiv f;
  f.mul = (uint32_t)((((uint64_t)1) << (31 + L)) / d + 1);
  f.sh = 31 + L;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
  return (uint32_t)(((uint64_t)n * f.mul) >> f.sh);
}

struct WgradKArgs {
  WgradArgs a;
  FastDiv dWo, dHo;
  int M, tiles, splits, items, pix_per_split;
  int ntg;  // tap groups of TPI taps
  int xcd;  // 1: workgroups of one XCD (blockIdx & 7) take consecutive items, so the tiles of one pixel split share its L2
  unsigned bytes_dy, bytes_x;
};

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef v2i32 __attribute__((address_space(3))) * lds_v2i32_ptr;
typedef unsigned char fp8_t;  // e4m3 operand bytes (ES == 1)
typedef short s16x8 __attribute__((ext_vector_type(8)));

// XOR applied to the 16-byte chunk index of a staged row (bf16 only; fp32 reads are conflict-free as they are)
template <int ES, int ROW_BYTES>
__device__ __forceinline__ int row_swz(int row) {
  if constexpr (ES == 4) return 0;
  else if constexpr (ES == 1) return ((row >> 1) & 3) << 1;  // 128-byte rows of bytes, 8-row transposed reads
  else if constexpr (ROW_BYTES == 256) return (row & 3) << 2;
  else return ((row >> 1) & 1) << 2;
}

template <typename T, int BMC, int BNC, int TPI>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradKArgs kp) {
  const WgradArgs& p = kp.a;
  constexpr int ES = (int)sizeof(T);
  constexpr int BKP = ES == 1 ? 64 : (ES == 2 ? MI355_WGRAD_BKP16 : 128 / ES);  // pixels per slab: 32 fp32, 64 bf16, 64 fp8
  constexpr int RB_A = BMC * ES;            // bytes of one staged dy row
  constexpr int RB_B = BNC * ES;            // bytes of one staged x row
  constexpr int A_BYTES = BKP * RB_A;
  constexpr int B_BYTES = BKP * RB_B;
  constexpr int STAGE = A_BYTES + TPI * B_BYTES;
  constexpr int LPR_A = RB_A / 16, LPR_B = RB_B / 16;  // lanes (16-byte chunks) per row
  constexpr int RPP_A = 64 / LPR_A, RPP_B = 64 / LPR_B;  // rows per 1 KiB piece
  constexpr int PW_A = A_BYTES / 4096, PW_B = B_BYTES / 4096;  // 1 KiB pieces per wave per slab
  constexpr int MI = BMC / 64, NI = BNC / 64;            // 32x32 tiles per wave (cout / cin)
  constexpr int NST = TPI * MI * NI * 4;                 // 16-byte stores per thread per item
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const int ckb = p.Ck / BNC;
  const i32x4 srdA = make_srd(p.dy, kp.bytes_dy);
  const i32x4 srdB = make_srd(p.x, kp.bytes_x);
  // item index of this workgroup in the round that starts at i - blockIdx.x (the ragged last round stays linear)
  auto remap = [&](int i) {
    const int r0 = i - (int)blockIdx.x;
    if (!kp.xcd || r0 + G > kp.items) return i;
    return r0 + (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
  };

  // ---- loader cursor (one slab ahead) -----------------------------------------------------------------------------
  // per-lane constants: row of each piece inside the slab and its swizzled 16-byte chunk
  int L_item = blockIdx.x, L_m = 0, L_mend = 0;
  int L_dh[TPI], L_dw[TPI];
  const unsigned pair_bytes = (unsigned)(p.pair_delta * ES);
  unsigned L_aoff = 0, L_boff = 0;  // uniform byte offsets: co0*ES / ci0*ES
  auto L_setup = [&]() {
    if (L_item >= kp.items) return;
    const int v = remap(L_item);
    const int tile = v % kp.tiles;
    const int split = v / kp.tiles;
    const int cib = tile % ckb;
    const int r1 = tile / ckb;
    const int t = r1 % kp.ntg;
    const int cb = r1 / kp.ntg;
#pragma unroll
    for (int jt = 0; jt < TPI; ++jt) {
      const Tap tp = p.taps[__builtin_amdgcn_readfirstlane(t * TPI + jt)];
      L_dh[jt] = tp.dh;
      L_dw[jt] = tp.dw;
    }
    L_aoff = (unsigned)(cb * BMC * ES);
    L_boff = (unsigned)(cib * BNC * ES);
    L_m = split * kp.pix_per_split;
    L_mend = L_m + kp.pix_per_split;
    if (L_mend > kp.M) L_mend = kp.M;
  };
  int S_m0 = 0, S_mend = 0;
  unsigned S_As = 0;
  auto L_begin = [&](int stage) {
    S_As = __builtin_amdgcn_readfirstlane(lds_addr(smem) + stage * STAGE);
    S_m0 = __builtin_amdgcn_readfirstlane(L_m);
    S_mend = __builtin_amdgcn_readfirstlane(L_mend);
  };
  auto L_piece = [&](int j) {  // j < PW_A: dy piece j;  else x piece (j-PW_A) % PW_B of tap (j-PW_A) / PW_B
    if (j < PW_A) {
      const int piece = wave * PW_A + j;
      const int row = piece * RPP_A + lane / LPR_A;
      const int c = (lane % LPR_A) ^ row_swz<ES, RB_A>(row);
      const int m = S_m0 + row;
      const unsigned off = (unsigned)m * (unsigned)(p.Cout * ES) + L_aoff + (unsigned)(c * 16);
      blds16(srdA, m < S_mend ? off : 0x80000000u, S_As + piece * 1024);
    } else {
      const int jt = (j - PW_A) / PW_B;
      const int piece = wave * PW_B + (j - PW_A) % PW_B;
      const int row = piece * RPP_B + lane / LPR_B;
      const int c = (lane % LPR_B) ^ row_swz<ES, RB_B>(row);
      const int m = S_m0 + row;
      const uint32_t q1 = fdiv((uint32_t)m, kp.dWo);
      const int ow = m - (int)q1 * p.Wo;
      const uint32_t n = fdiv(q1, kp.dHo);
      const int oh = (int)q1 - (int)n * p.Ho;
      const int ih = oh * p.IS + L_dh[jt];
      const int iw = ow * p.IS + L_dw[jt];
      const bool ok = m < S_mend && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      // paired taps (stem): the upper half of the 64 staged "channels" comes from the next image row
      const unsigned pair = c * (16 / ES) >= 32 ? pair_bytes : 0u;
      const unsigned off = (unsigned)(((int)n * p.Hin + ih) * p.Win + iw) * (unsigned)(p.pix_stride * ES) + L_boff +
                           (unsigned)(c * 16) + pair;
      blds16(srdB, ok ? off : 0x80000000u, S_As + A_BYTES + jt * B_BYTES + piece * 1024);
    }
  };
  auto L_advance = [&]() {
    L_m += BKP;
    if (L_m < L_mend) return;
    L_item += G;
    L_setup();
  };
  constexpr int NPC = PW_A + TPI * PW_B;  // pieces per wave per slab

  // ---- per-lane fragment coordinates -------------------------------------------------------------------------------
  const int cobase = wm * (BMC / 2);  // + mi*32 : output channels, on the lanes of D
  const int cibase = wn * (BNC / 2);  // + ni*32 : input channels, in the registers of D
  const int hh = lane >> 5, c31 = lane & 31;
  // bf16 transposed reads: group g = lane>>4 reads k rows 8*(g>>1)+q (+4), columns 16*(g&1) + 4*pp
  const int tg = lane >> 4, tw = lane & 15, tq = tw >> 2, tpp = tw & 3;
  const int t_krow = 8 * (tg >> 1) + tq;
  const int t_col = 16 * (tg & 1) + 4 * tpp;  // element column inside the 32-wide tile
  // fp8 transposed reads: group g reads k rows 8*(g>>1) .. +7 (lane pair q = tw>>1 supplies row q), byte columns 16*(g&1) + 8*(tw&1)
  const int t8_krow = 8 * (tg >> 1) + (tw >> 1);
  const int t8_col = 16 * (tg & 1) + 8 * (tw & 1);

  L_setup();
  int stage = 0;
  if (L_item < kp.items) {
    L_begin(0);
#pragma unroll
    for (int j = 0; j < NPC; ++j) L_piece(j);
    L_advance();
  }
  int pending_st = 0;

  for (int item = blockIdx.x; item < kp.items; item += G) {
    const int v = remap(item);
    const int tile = v % kp.tiles;
    const int split = v / kp.tiles;
    const int cib = tile % ckb;
    const int r1 = tile / ckb;
    const int t = r1 % kp.ntg;
    const int cb = r1 / kp.ntg;
    const int mbeg = split * kp.pix_per_split;
    int mend = mbeg + kp.pix_per_split;
    if (mend > kp.M) mend = kp.M;

    f32x16 acc[TPI][MI][NI];
#pragma unroll
    for (int jt = 0; jt < TPI; ++jt)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[jt][mi][ni][r] = 0.f;

    for (int m0 = mbeg; m0 < mend; m0 += BKP) {
      if (pending_st == NST) {
        if constexpr (NST == 16) MI355_WAIT_VM(16);
        else if constexpr (NST == 12) MI355_WAIT_VM(12);
        else if constexpr (NST == 8) MI355_WAIT_VM(8);
        else MI355_WAIT_VM(4);
      } else {
        MI355_WAIT_VM(0);
      }
      pending_st = 0;
      MI355_LDS_BARRIER();
      const bool lv = L_item < kp.items;
      if (lv) L_begin(stage ^ 1);
      const char* Ad = smem + stage * STAGE;
      const char* Bx = Ad + A_BYTES;
      // the next slab's NPC pieces are spread evenly over the NS inner steps of this one
      if constexpr (ES == 4) {
        constexpr int NS = BKP / 2;
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) {
          const int k = 2 * kk + hh;
          float dv[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) dv[mi] = *reinterpret_cast<const float*>(Ad + k * RB_A + (cobase + mi * 32 + c31) * 4);
          if (lv) {
#pragma unroll
            for (int j = kk * NPC / NS; j < (kk + 1) * NPC / NS; ++j) L_piece(j);
          }
#pragma unroll
          for (int jt = 0; jt < TPI; ++jt) {
            float xv[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              xv[ni] = *reinterpret_cast<const float*>(Bx + jt * B_BYTES + k * RB_B + (cibase + ni * 32 + c31) * 4);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[jt][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[ni], dv[mi], acc[jt][mi][ni], 0, 0, 0);
          }
        }
      } else if constexpr (ES == 1) {
        // The four k-steps of 16 pixels of a slab as ONE v_mfma_f32_32x32x64_f8f6f4 per accumulator tile: the 4 x 8 bytes a lane reads (transposed) for them are the
        // 32-byte operand of the K = 64 form — the same bytes of dy and of x on both sides, so which pixel a byte belongs to need not be known (the pixel is
        // a summation index) — at twice the rate of four 32x32x16 fp8 instructions (tools/micro/mfma_rate.hip).
        static_assert(BKP == 64, "one K = 64 instruction per slab");
        typedef int i32x8v __attribute__((ext_vector_type(8)));
        i32x8v df[MI];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k0 = q * 16 + t8_krow;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int byte = cobase + mi * 32 + t8_col;
            const char* ap = Ad + k0 * RB_A + ((((byte >> 4) ^ row_swz<ES, RB_A>(k0)) << 4) | (byte & 15));
            const auto v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i32_ptr)(ap));
            df[mi][2 * q] = v[0];
            df[mi][2 * q + 1] = v[1];
          }
          if (lv) {
#pragma unroll
            for (int j = q * NPC / 4; j < (q + 1) * NPC / 4; ++j) L_piece(j);
          }
        }
#pragma unroll
        for (int jt = 0; jt < TPI; ++jt) {
          i32x8v xf[NI];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k0 = q * 16 + t8_krow;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int byte = cibase + ni * 32 + t8_col;
              const char* ap = Bx + jt * B_BYTES + k0 * RB_B + ((((byte >> 4) ^ row_swz<ES, RB_B>(k0)) << 4) | (byte & 15));
              const auto v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i32_ptr)(ap));
              xf[ni][2 * q] = v[0];
              xf[ni][2 * q + 1] = v[1];
            }
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              auto c = acc[jt][mi][ni];
              asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(xf[ni]), "v"(df[mi]));
              acc[jt][mi][ni] = c;
            }
        }
      } else {
        constexpr int NS = BKP / 16;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          const int k0 = ks * 16 + t_krow;  // rows k0 and k0+4 share (row&3) and ((row>>1)&1): same swizzle
          bf16x8 df[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int byte = (cobase + mi * 32 + t_col) * 2;
            const char* ap = Ad + k0 * RB_A + ((((byte >> 4) ^ row_swz<ES, RB_A>(k0)) << 4) | (byte & 15));
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * RB_A));
            const s16x8 tmp = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            df[mi] = __builtin_bit_cast(bf16x8, tmp);
          }
          if (lv) {  // the next slab's pieces go out between this k-step's LDS reads and its MFMAs
#pragma unroll
            for (int j = ks * NPC / NS; j < (ks + 1) * NPC / NS; ++j) L_piece(j);
          }
#pragma unroll
          for (int jt = 0; jt < TPI; ++jt) {
            bf16x8 xf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int byte = (cibase + ni * 32 + t_col) * 2;
              const char* ap = Bx + jt * B_BYTES + k0 * RB_B + ((((byte >> 4) ^ row_swz<ES, RB_B>(k0)) << 4) | (byte & 15));
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * RB_B));
              const s16x8 tmp = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
              xf[ni] = __builtin_bit_cast(bf16x8, tmp);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[jt][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ni], df[mi], acc[jt][mi][ni], 0, 0, 0);
          }
        }
      }
      if (lv) L_advance();
      stage ^= 1;
    }

    // (the e4m3 path issues its matrix instructions as text: the compiler does not know their result latency, so the wait states in front of the first
    //  read of an accumulator are spelled out)
      if constexpr (ES == 1) {
        __builtin_amdgcn_sched_barrier(0);  // (nothing that reads an accumulator may be scheduled above the wait states)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    // partial[split][co][wtap][ci]: D rows (registers) = input channels, D columns (lanes) = output channels
    float* outp = p.partial + (size_t)split * p.Cout * p.wtaps * p.Ck;
#pragma unroll
    for (int jt = 0; jt < TPI; ++jt) {
      const int wtap = p.taps[__builtin_amdgcn_readfirstlane(t * TPI + jt)].wtap;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int co = cb * BMC + cobase + mi * 32 + c31;
        float* rowp = outp + ((size_t)co * p.wtaps + wtap) * p.Ck + cib * BNC + cibase;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[jt][mi][ni][4 * g], acc[jt][mi][ni][4 * g + 1], acc[jt][mi][ni][4 * g + 2],
                             acc[jt][mi][ni][4 * g + 3]};
            *reinterpret_cast<f32x4*>(rowp + ni * 32 + 8 * g + 4 * hh) = v;
          }
      }
    }
    pending_st += NST;
  }
}

// dst[i] = beta*dst[i] + sum_k partial[k][i].  A block is (256/SG) 16-byte columns x SG split groups: group g adds
// slabs g, g+SG, ... in order (4 loads in flight), then the groups are added in order 0..SG-1 through LDS — the
// summation order depends only on (splits, SG), never on timing.  SG > 1 is for small outputs, where one thread per
// column would walk hundreds of slabs alone.
template <int SG>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* partial, int splits, size_t stride,
                                                            float* dst, size_t n4, float beta, const float* sa, const float* sb) {
  constexpr int COLS = 256 / SG;
  const float oscale = sa ? 1.f / (sa[0] * sb[0]) : 1.f;  // fp8 wgrad: the sums are in units of the two operands' scales
  __shared__ f32x4 red[SG > 1 ? SG : 1][COLS];
  const int col = threadIdx.x % COLS, sg = threadIdx.x / COLS;
  for (size_t i0 = (size_t)blockIdx.x * COLS; i0 < n4; i0 += (size_t)gridDim.x * COLS) {
    const size_t i = i0 + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
      int k = sg;
      for (; k + 3 * SG < splits; k += 4 * SG) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(partial + (size_t)k * stride + i * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(partial + (size_t)(k + SG) * stride + i * 4);
        const f32x4 c = *reinterpret_cast<const f32x4*>(partial + (size_t)(k + 2 * SG) * stride + i * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(partial + (size_t)(k + 3 * SG) * stride + i * 4);
        s += a;
        s += b;
        s += c;
        s += d;
      }
      for (; k < splits; k += SG) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * stride + i * 4);
    }
    if constexpr (SG > 1) {
      red[sg][col] = s;
      __syncthreads();
      if (sg == 0) {
#pragma unroll
        for (int g = 1; g < SG; ++g) s += red[g][col];
      }
    }
    if (sg == 0 && i < n4) {
      f32x4* d = reinterpret_cast<f32x4*>(dst + i * 4);
      if (sa) s *= oscale;
      if (beta != 0.f) s += beta * (*d);
      *d = s;
    }
    if constexpr (SG > 1) __syncthreads();
  }
}

// dw[64][7][7][3] <- partial[s][64][4][64]: tap pair kh>>1, element (kh&1)*32 + kw*4 + c
__global__ void stem_unpack_kernel(const float* __restrict__ partial, int splits, float* __restrict__ dw, float beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 7 * 7 * 3) return;
  const int c = i % 3;
  int t = i / 3;
  const int kw = t % 7;
  t /= 7;
  const int kh = t % 7;
  const int co = t / 7;
  const size_t src = ((size_t)co * 4 + (kh >> 1)) * 64 + (kh & 1) * 32 + kw * 4 + c;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(size_t)k * 64 * 4 * 64 + src];
  dw[i] = (beta != 0.f ? beta * dw[i] : 0.f) + s;
}

inline int wg_bmc(int Cout) { return Cout % 128 == 0 ? 128 : 64; }
inline int wg_bnc(int Ck) { return Ck % 128 == 0 ? 128 : 64; }
// taps per item: only the 64x64 tile has the registers (TPI*16 accumulators) and needs it (fetch-bound otherwise)
inline int wg_tpi(int Cout, int Ck, int ntaps) {
  if (wg_bmc(Cout) != 64 || wg_bnc(Ck) != 64) return 1;
  if (ntaps % 4 == 0) return 4;
  if (ntaps % 3 == 0) return 3;
  return 1;
}

template <typename T, int BMC, int BNC, int TPI>
int launch_t(const WgradArgs& a, int splits, hipStream_t stream) {
  constexpr int ES = (int)sizeof(T);
  constexpr int BKP = ES == 1 ? 64 : (ES == 2 ? MI355_WGRAD_BKP16 : 128 / ES);
  WgradKArgs k;
  k.a = a;
  k.M = a.N * a.Ho * a.Wo;
  k.dWo = make_fastdiv((uint32_t)a.Wo);
  k.dHo = make_fastdiv((uint32_t)a.Ho);
  k.ntg = a.ntaps / TPI;
  k.tiles = (a.Cout / BMC) * k.ntg * (a.Ck / BNC);
  k.splits = splits;
  k.items = k.tiles * splits;
  // splits are planned in 64-pixel units so that fp32 (32-pixel slabs) and bf16 (64) cut the pixels identically
  k.pix_per_split = cdiv(cdiv(k.M, 64), splits) * 64;
  static_assert(64 % BKP == 0, "slab size must divide the planning unit");
  const size_t bytes_dy = (size_t)k.M * a.Cout * ES;
  const size_t bytes_x = (size_t)a.N * a.Hin * a.Win * a.pix_stride * ES;
  MI355_ARG(bytes_dy < 0x80000000ull && bytes_x < 0x80000000ull, "wgrad: tensor exceeds the 2 GiB buffer-offset range");
  k.bytes_dy = (unsigned)bytes_dy;
  k.bytes_x = (unsigned)bytes_x;
#ifdef MI355_PROBES
  if (const int dbg = probe_env("MI355_WGRAD_DBG")) {  // timing probes (profiling build; results are wrong): zero-record descriptors = no load traffic
    if (dbg & 1) k.bytes_dy = 0;
    if (dbg & 2) k.bytes_x = 0;
  }
#endif
  const int grid = k.items < MAX_WG ? k.items : MAX_WG;
  // every tile of a pixel split fetches the same dy / x slabs: keep the tiles of a split on one XCD (one L2) wherever the
  // grid divides over the 8 XCDs.  Same-box whole-step A/B at batch 256 (tools/ab_step.sh MI355_WGRAD_XCD): 20.46 -> 20.36 ms
  // with it on for every layer (round 1 had it for <= 16 tiles only, from warm-cache per-op timings where the MALL hides
  // the re-fetch).
  k.xcd = (k.tiles > 1 && grid % 8 == 0) ? 1 : 0;
  const size_t lds = (size_t)2 * BKP * (BMC + TPI * BNC) * ES;
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in
    MI355_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<T, BMC, BNC, TPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<T, BMC, BNC, TPI>), dim3(grid), dim3(256), lds, stream, k);
  MI355_LAUNCH_CHECK();
  note_kernel("wgrad<%s,%d,%d,%d>", sizeof(T) == 4 ? "f32" : (sizeof(T) == 2 ? "bf16" : "e4m3"), BMC, BNC, TPI);
  return 0;
}

template <typename T>
int launch_d(const WgradArgs& a, int splits, hipStream_t stream) {
  const int bmc = wg_bmc(a.Cout), bnc = wg_bnc(a.Ck);
  if (bmc == 128) return bnc == 128 ? launch_t<T, 128, 128, 1>(a, splits, stream) : launch_t<T, 128, 64, 1>(a, splits, stream);
  if (bnc == 128) return launch_t<T, 64, 128, 1>(a, splits, stream);
  switch (wg_tpi(a.Cout, a.Ck, a.ntaps)) {
    case 4: return launch_t<T, 64, 64, 4>(a, splits, stream);
    case 3: return launch_t<T, 64, 64, 3>(a, splits, stream);
    default: return launch_t<T, 64, 64, 1>(a, splits, stream);
  }
}

}  // namespace

int plan_wgrad_splits(int dtype, int M, int Cout, int ntaps, int Ck) {
  const int tiles = (Cout / wg_bmc(Cout)) * (ntaps / wg_tpi(Cout, Ck, ntaps)) * (Ck / wg_bnc(Ck));
  const int chunks = cdiv(M, 64);
  // The launch runs ceil(tiles*splits / MAX_WG) rounds of items of cdiv(chunks, splits) 64-pixel chunks each (+ ~2
  // chunks worth of prologue/epilogue per item): take the split count that minimises rounds x item length.  fp32 is
  // allowed two rounds (compute-bound: the finer grain evens out the tail; measured same-box A/B), bf16 one (the fp32
  // partial slabs are what bounds the small layers there); keep >= 8 chunks per item so the pipeline has a body.
  const int max_rounds = dtype == MI355_F32 ? 2 : 1;
  int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;
  if (max_splits > max_rounds * MAX_WG) max_splits = max_rounds * MAX_WG;
  int best = 1;
  long best_cost = -1;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const int rounds = cdiv(tiles * sp, MAX_WG);
    if (rounds > max_rounds && sp > 1) break;
    const long cost = (long)rounds * (cdiv(chunks, sp) + 2);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = sp;
    }
  }
  const int cps = cdiv(chunks, best);
  return cdiv(chunks, cps);
}

int plan_wgrad(int dtype, const WgradArgs& a) {
  const int g = wg3_plan(dtype, a);
  return g > 0 ? g : plan_wgrad_splits(dtype, a.N * a.Ho * a.Wo, a.Cout, a.ntaps, a.Ck);
}

int launch_wgrad(int dtype, const WgradArgs& a, int splits, hipStream_t stream) {
  MI355_ARG(a.dy && a.x && a.partial, "wgrad: null pointer");
  MI355_ARG(!knobs().error[0], "%s", knobs().error);
  // The split count is a PLAN made earlier (context creation, or the per-op call a moment ago) from process-wide state — reserved CUs,
  // the kernel switches.  If that state changed in between (the tile tests flip MI355_IGEMM8 between two backward calls of one context),
  // the generated kernel the plan was made for may no longer be the one this launch takes: the implicit-GEMM kernel then runs with the
  // planned count, which is fine as long as every split still has pixels — a split without any would leave its slab unwritten and
  // splitk_reduce would add stale memory.  That case is refused.  (fp8 twins always take the implicit-GEMM plan.)
  if (dtype != MI355_FP8 && splits > 0 && wg3_plan(dtype, a) == splits) return launch_wg3(a, splits, stream);
  {
    const long M = (long)a.N * a.Ho * a.Wo;
    const long pps = (long)cdiv(cdiv((int)M, 64), splits > 0 ? splits : 1) * 64;
    MI355_ARG(splits >= 1 && (long)(splits - 1) * pps < M, "wgrad: %d splits of %ld pixels leave a split without pixels (M = %ld): the plan is stale", splits, pps, M);
  }
  MI355_ARG(a.Cout % 64 == 0 && a.Ck % 64 == 0, "wgrad: Cout=%d Ck=%d must be multiples of 64", a.Cout, a.Ck);
  MI355_ARG(splits >= 1, "wgrad: splits=%d", splits);
  MI355_ARG(((size_t)a.pix_stride * dtype_size(dtype)) % 8 == 0, "wgrad: pixel stride not 8-byte aligned");
  if (dtype == MI355_F32) return launch_d<float>(a, splits, stream);
  if (dtype == MI355_BF16) return launch_d<bf16_t>(a, splits, stream);
  if (dtype == MI355_FP8) {  // e4m3 twins of dy and x: the caller scales the result by 1 / (scale_dy * scale_x) (splitk_reduce)
    MI355_ARG(a.Cout % 128 == 0 && a.Ck % 128 == 0 && a.pair_delta == 0, "wgrad fp8: Cout=%d Ck=%d must be multiples of 128", a.Cout, a.Ck);
    return launch_t<fp8_t, 128, 128, 1>(a, splits, stream);
  }
  set_error("wgrad: bad dtype %d", dtype);
  return MI355_E_ARG;
}

int launch_splitk_reduce(const float* partial, int splits, size_t stride, float* dst, size_t n, float beta,
                         hipStream_t stream, const float* sa, const float* sb) {
  MI355_ARG(n % 4 == 0 && stride % 4 == 0, "splitk_reduce: n=%zu stride=%zu must be multiples of 4", n, stride);
  const size_t n4 = n / 4;
  // small outputs with many slabs: several threads per column (see the kernel)
  const int sg = (splits >= 64 && n4 < 16384) ? 16 : (splits >= 16 && n4 < 65536) ? 4 : 1;
  const int cols = 256 / sg;
  int blocks = (int)((n4 + cols - 1) / cols);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (sg == 16)
    hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(blocks), dim3(256), 0, stream, partial, splits, stride, dst, n4, beta, sa, sb);
  else if (sg == 4)
    hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, stream, partial, splits, stride, dst, n4, beta, sa, sb);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, stream, partial, splits, stride, dst, n4, beta, sa, sb);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_stem_unpack(float* partial, int splits, float* dw, float beta, hipStream_t stream) {
  const int n = 64 * 7 * 7 * 3;
  if (splits > 1) {  // slab 0 <- sum of the slabs (a column is read and written by the same threads: in place is safe)
    MI355_TRY(launch_splitk_reduce(partial, splits, (size_t)64 * 4 * 64, partial, (size_t)64 * 4 * 64, 0.f, stream, nullptr, nullptr));
    splits = 1;
  }
  hipLaunchKernelGGL(stem_unpack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, partial, splits, dw, beta);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace mi355
