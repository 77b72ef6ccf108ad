// conv_igemm8.hip — bf16 implicit-GEMM convolution forward / data-gradient, 8-wave ping-pong form, for gfx950 (MI355X).
//
// Same contract as conv_igemm.hip (IgemmArgs in common.h: forward, dgrad through tap classes, optional addend under a
// ReLU bit mask, BN statistics / BN-backward sums in the epilogue), i.e. the cuDNN conv fwd / dgrad work under
// `model(data)` and `loss.backward()` of the reference (call form sota_imagenet/callbacks.py:316-317).  launch_igemm()
// sends the long-reduction bf16 launches here; the short, output-heavy ones stay on the 2-workgroups-per-CU kernel.
//
//   workgroup   ONE per CU, 8 waves as 2 (pixel rows) x 4 (channels); a wave owns BM/2 pixels x BN/4 channels in
//               16x16 accumulator tiles (v_mfma_f32_16x16x32_bf16; operands swapped, D^T = W * A^T, so a lane ends with
//               4 consecutive channels of one pixel).  Tiles: 256|224 pixels x 256|128 channels, K in 128-byte slabs
//               (one tap x 64 channels).  224-pixel tiles exist because every layer of the network has 49 * 2^k * N
//               pixels: 224 | 50176, so the tile count fills 7/8 (or all) of the 256 CUs where 256-pixel tiles fill 49/64.
//   phases      a k-tile is cut into NPH = BN/64 phases of <= 16 MFMAs per wave (one quadrant of the wave's tile):
//                 { fragment ds_read_b128s of the quadrant | LDS-DMA pieces of a LATER k-tile | counted vmcnt }
//                 s_barrier, lgkmcnt(0), MFMAs, s_barrier
//               The two wave rows run one barrier apart (wave row 1 enters through an extra barrier), so on every SIMD
//               one wave is in its MFMA section while its partner reads fragments and issues loads.
//   staging     buffer_load_dwordx4 ... offen lds, 1 KiB per wave-instruction, per-lane gathered source offsets
//               (out-of-image rows: out-of-range offset -> the range check returns zeros); LDS image [row][128 B] with
//               the 16-byte chunk index XORed with (row >> 1) & 7 on the source offset and on the fragment read
//               (conflict-free for the 16x16x32 operand read).  A k-tile is staged in NPH groups
//               (A rows of quadrant half 0, weight rows of half 0, weight rows of half 1, A rows of half 1) and group
//               g is issued D = NSTG*NPH - 2 phases before the phase that first reads it: the region it overwrites was
//               last read two phases earlier (WAR across the staggered wave rows needs two), and the wait that retires
//               it sits one phase before its first read (RAW needs the wait, then a barrier every wave has passed).
//   persistent  a workgroup walks (row tile x n-tile group, n-tile, k-tile) as one stream; the loader cursor runs D
//               groups ahead of the MFMAs, also across tile boundaries, so the next tile's first k-tiles land under the
//               epilogue.  Epilogue stores stay in flight behind counted waits (vmcnt counts them too).
//   epilogue    straight from the accumulators: lanes l and l^16 exchange halves (ds_swizzle SWAP16) so that every lane
//               holds 8 consecutive channels of one pixel = one 16-byte vector of the output, the unit the addend, the
//               ReLU bit masks and the BN tensors are addressed in; a wave-instruction writes 64 contiguous bytes of 16
//               pixels.  Statistics: per-lane sums over the wave's pixels, a fixed-order DPP row reduction, ONE lane per
//               (wave, channel) adds into the workgroup's LDS accumulator, one flush per workgroup — no atomics.
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>

#include "common.h"
#include "lds_dma.h"
#include "vec.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
namespace mi355 {

namespace {

// > 64 KiB of dynamic LDS needs an opt-in per kernel symbol (once)
void lds_opt_in8(const void* fn, size_t lds) {
  if (lds <= 64 * 1024) return;
  static std::mutex mu;
  static std::set<const void*> done;
  std::lock_guard<std::mutex> g(mu);
  if (done.count(fn)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.insert(fn);
}

__device__ __attribute__((aligned(256))) unsigned char g8_trash[512 * 16];

struct Igemm8KArgs {
  IgemmArgs a;
  int mtiles, ngroups, ntpg, items;
  unsigned bytes_in, bytes_wt;
  unsigned magW, magHW;  // floor(2^32 / d) + 1 for d = Wsub, Hsub * Wsub (exact quotients while m * d < 2^32)
  int HW;
  float oscale;          // fp8 operands: 1 / (scale_x * scale_w), applied to the accumulators
  const float* sc_in;    // fp8 operands, executor path: per-tensor scales in device memory (the accumulators are multiplied
  const float* sc_wt;    // by oscale / (*sc_in * *sc_wt)); null: oscale alone
  int korder;            // 0: taps outer, channel chunks inner;  1: channel chunks outer, taps inner (A re-reads stay close)
  int xcd;               // 1: workgroup b walks the items of virtual workgroup xcd_rank(b): the workgroups resident on one XCD
                         // (hardware round-robin, b % 8) take CONSECUTIVE items = the n-tiles of the same row tiles
};

// -DMI355_STAMP8: cycle stamps of waves 0 and 4 of workgroup 0 (one of each wave row), 4 per phase, parked in LDS and
// copied out at the end; read back by tools/stamp8.py (profiling builds only — the stamps' lgkmcnt(0) changes the overlap)
// -DMI355_ITEMSTAMP: 100 MHz realtime stamps per workgroup and item (k-loop start, k-loop end, epilogue end) of wave 0,
// read back by tools/items8.py (profiling builds only)
#ifdef MI355_ITEMSTAMP
__device__ unsigned long long g8_items[512 * 16 * 4];
#define ITEMSTAMP(slot)                                                                                           \
  do {                                                                                                            \
    if (tid == 0 && item_n < 16) g8_items[((int)blockIdx.x * 16 + item_n) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define ITEMSTAMP(slot)
#endif
#ifdef MI355_STAMP8
__device__ unsigned long long g8_stamps[2 * 1024];
#define STAMP8(slot)                                                                                         \
  do {                                                                                                       \
    const unsigned long long t_ = __builtin_readcyclecounter();                                              \
    if (blockIdx.x == 0 && lane == 0 && wn == 0 && stamp_n < 1024 / 4)                                       \
      reinterpret_cast<unsigned long long*>(smem + STAMP_OFF)[(wr * 1024) + stamp_n * 4 + (slot)] = t_;      \
  } while (0)
#else
#define STAMP8(slot)
#endif

typedef long i64x2 __attribute__((ext_vector_type(2)));

constexpr std::integral_constant<int, 0> I0{};
constexpr std::integral_constant<int, 1> I1{};

template <int N>
__device__ __forceinline__ void wait_vm8() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CTRL>
__device__ __forceinline__ float row_shr_add(float x) {  // x + (x of the lane CTRL-0x110 to the left in the 16-lane row, 0 beyond)
  const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true);
  return x + __int_as_float(y);
}
__device__ __forceinline__ float row_sum16(float x) {  // lane 15 of every 16-lane row ends with the row's sum (fixed order)
  x = row_shr_add<0x111>(x);
  x = row_shr_add<0x112>(x);
  x = row_shr_add<0x114>(x);
  x = row_shr_add<0x118>(x);
  return x;
}

// EB: bytes per operand element — 2: bf16;  1: OCP fp8 e4m3 operands (v_mfma_f32_16x16x32_fp8_fp8, same matrix rate, half
// the operand bytes: a 128-byte LDS row holds 128 channels and one ds_read_b128 feeds TWO MFMAs), bf16 output scaled by oscale
template <int BM, int BN, int STATS, int FAT, int EB = 2>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const Igemm8KArgs kp) {
  const IgemmArgs& p = kp.a;
  // FAT: phases of up to 32 MFMAs per wave (2 per k-tile at BN 256, 1 at BN 128) — half the barriers of the 16-MFMA
  // phases, and a load section (fragment reads + LDS-DMA issue) fits under the partner's MFMA section
  constexpr int NPH = BN / 64;                  // 16-MFMA phases (= staging groups of the slim form) per k-tile
  constexpr int NSTG = BN == 256 ? 2 : 3;       // LDS ring stages
  constexpr int D = NPH * (NSTG - 1);           // staging lookahead in groups: the loader runs NSTG - 1 k-tiles ahead
  constexpr int TM = BM / 2;                    // pixels per wave row
  constexpr int MT = TM / 16;                   // 16-pixel tiles per wave: 8 | 7
  constexpr int MT0 = 4, MT1 = MT - MT0;        // ... of the two quadrant halves
  constexpr int TN = BN / 4;                    // channels per wave: 64 | 32
  constexpr int NT = TN / 16;                   // 16-channel tiles per wave: 4 | 2
  constexpr int A_BYTES = BM * 128;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int RING = NSTG * STAGE;
  constexpr int GL = NPH == 4 ? 2 : 3;          // LDS-DMA pieces per wave per group
  constexpr int WAITN = FAT ? (BN == 256 ? 8 : 6) : 2 * GL;  // pieces younger than the group a wait retires
  constexpr int NST = MT * NT / 2;              // 16-byte stores per thread per tile
  constexpr int BK = 128 / EB;                  // elements of K per k-tile (one 128-byte LDS row)
  static_assert(BM == 256 || BM == 224, "BM");
  static_assert(BN == 256 || BN == 128, "BN");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [ring][1 KiB sink of the padding pieces][stat_acc: 2 wave rows x channels of this workgroup's n-tiles x {sum, sum2}]
  constexpr int SINK = RING;
  float* stat_acc = reinterpret_cast<float*>(smem + RING + 1024);
#ifdef MI355_STAMP8
  constexpr int STAMP_OFF = RING + 1024;  // (profiling builds run without statistics)
  int stamp_n = 0;
  for (int i = threadIdx.x; i < 2 * 1024; i += 512) reinterpret_cast<unsigned long long*>(smem + STAMP_OFF)[i] = 0ull;
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wn = wave & 3;
  const int G = gridDim.x;
  // XCD-aware item order: virtual index = (workgroups on lower-numbered XCDs) + (rank among this XCD's workgroups)
  const int bx = __builtin_amdgcn_readfirstlane(
      kp.xcd ? (int)(blockIdx.x & 7u) * (G >> 3) + min((int)(blockIdx.x & 7u), G & 7) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  const int Msub = p.N * p.Hsub * p.Wsub;
  const int kcpt = p.Ck / BK;  // k-tiles per tap

  float oscale = kp.oscale;
  if constexpr (EB == 1) {
    if (kp.sc_in) oscale = kp.oscale / (kp.sc_in[0] * kp.sc_wt[0]);  // uniform: two scalar loads
  }
  const i32x4 srdA = make_srd(p.in, kp.bytes_in);
  const i32x4 srdB = make_srd(p.wt, kp.bytes_wt);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));

  // m -> (n, i, j) of the sub-grid
  auto decode = [&](int m, int& n, int& i, int& j) __attribute__((always_inline)) {
    n = (int)__umulhi((unsigned)m, kp.magHW);
    const int rem = m - n * kp.HW;
    i = (int)__umulhi((unsigned)rem, kp.magW);
    j = rem - i * p.Wsub;
  };

  // ---- loader ------------------------------------------------------------------------------------------------------
  // Every wave issues GL pieces per group.  A piece = 8 LDS rows; lane -> row prow of the piece, physical chunk pch.
  // This wave's pieces of a quarter are j = 2*wave + i (i = 0, 1):
  //   A quarter h: wave row j >> 3 = wave >> 2, LDS rows (wave >> 2)*TM + h*64 + 8*(j & 7) ... + 8
  //   B quarter h (NPH 4): rows (wave >> 1)*64 + h*32 + 8*(j & 3) ... + 8;   NPH 2, half h: rows h*64 + 8*wave ... + 8
  // so the LDS destination of a piece is a per-wave base plus a compile-time constant, and (row >> 1) & 7 of this lane's
  // row is (prow >> 1) | (i << 2) — TM, 64, 32 and 16 are multiples of 16 (NPH 2: (prow >> 1) | ((wave & 1) << 2)).
  const int prow = lane >> 3, pch = lane & 7;
  const unsigned ldsA0 = (unsigned)(((wave >> 2) * TM + 8 * ((2 * wave) & 7)) * 128);
  const unsigned ldsB0 = (unsigned)(A_BYTES + (NPH == 4 ? ((wave >> 1) * 64 + 8 * ((2 * wave) & 3)) : 8 * wave) * 128);
  constexpr int A_STEP_I = 1024, A_STEP_H = 64 * 128;
  constexpr int B_STEP_I = 1024, B_STEP_H = (NPH == 4 ? 32 : 64) * 128;
  // padding pieces (TM = 112: the second quarter has 48 rows per wave row): wave-uniform, h = 1 only
  const bool a_pad0 = TM != 128 && 8 * ((2 * wave) & 7) >= TM - 64;
  const bool a_pad1 = TM != 128 && 8 * ((2 * wave + 1) & 7) >= TM - 64;
  unsigned a_off[4];  // [h*2 + i]: byte offset of the row's pixel at tap offset (0,0) + this lane's (swizzled) 16-byte chunk
  unsigned a_inv[4];  // bit t set: tap t of the unit's class falls outside the image for this row (all set: no such row)
  unsigned b_off[2];  // [i]: this lane's weight row of piece i of quarter 0, n-tile 0, tap 0 (+ swizzled chunk)
  const unsigned w_row = (unsigned)(p.wtaps * p.Ck * EB);  // bytes of one weight row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, sw;
    if constexpr (NPH == 4) {
      row = (wave >> 1) * 64 + 8 * (((2 * wave) & 3) + i);
      sw = (prow >> 1) | (i << 2);
    } else {
      row = 8 * wave;  // (one piece per half: i unused)
      sw = (prow >> 1) | ((wave & 1) << 2);
    }
    b_off[i] = (unsigned)(row + prow) * w_row + (unsigned)((pch ^ sw) * 16);
  }
  // Per-k-tile deltas of the loader's class, k-tile i in lane i & 63 of table i >> 6 (<= 128 k-tiles per class): read back
  // with v_readlane, so moving the cursor costs no memory access and no (tap, chunk) arithmetic.
  //   tabA: byte delta of the A source = ((dh*Win + dw)*pix_stride + chunk*64) * 2
  //   tabB: byte delta of the weight source = (wtap*Ck + chunk*64) * 2 (a multiple of 128), | (31 - tap) in the low bits
  int v_tabA0 = 0, v_tabA1 = 0, v_tabB0 = 0, v_tabB1 = 0;

  // cursor (all wave-uniform): unit L_seq, its class / n-tile group, n-tile, k-tile, ring stage
  int L_seq = 0, L_cls = -1, L_grp = 0, L_nt = 0, L_kt = 0, L_kend = 0, L_stage = 0;
  bool L_valid = true;
  // uniform state of the loader's CURRENT k-tile (set by L_tile whenever the cursor moves)
  int S_dA = 0;
  unsigned S_dB = 0, S_nB = 0, S_baseA = 0, S_baseB = 0, S_sh = 0;
  auto L_tile = [&]() __attribute__((always_inline)) {
    const int kt = __builtin_amdgcn_readfirstlane(L_kt);
    const int k6 = kt & 63;
    const int a0 = __builtin_amdgcn_readlane(v_tabA0, k6), a1 = __builtin_amdgcn_readlane(v_tabA1, k6);
    const int b0 = __builtin_amdgcn_readlane(v_tabB0, k6), b1 = __builtin_amdgcn_readlane(v_tabB1, k6);
    const int ea = kt < 64 ? a0 : a1, eb = kt < 64 ? b0 : b1;
    S_dA = ea;
    S_sh = (unsigned)eb & 31u;
    S_dB = S_nB + ((unsigned)eb & ~127u);
    const unsigned sb = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(L_stage) * STAGE;
    S_baseA = sb + ldsA0;
    S_baseB = sb + ldsB0;
  };
  auto L_ntile = [&]() __attribute__((always_inline)) {  // weight rows of the loader's n-tile
    S_nB = (unsigned)__builtin_amdgcn_readfirstlane((L_grp * kp.ntpg + L_nt) * BN) * w_row;
  };
  auto L_setup = [&]() __attribute__((always_inline)) {  // decode the rows this lane stages for unit L_seq; skips units without taps
    for (;;) {
      // (the cursor is wave-uniform by construction; the readfirstlanes say so to hipcc, whose divergence analysis
      // otherwise turns the cursor into VGPR values under exec masks)
      const int item = __builtin_amdgcn_readfirstlane(bx + L_seq * G);
      L_valid = item < kp.items;
      if (!L_valid) return;
      const int rowtile = item / kp.ngroups;
      L_grp = __builtin_amdgcn_readfirstlane(item - rowtile * kp.ngroups);
      const int cls = __builtin_amdgcn_readfirstlane(rowtile / kp.mtiles);
      const int mt = __builtin_amdgcn_readfirstlane(rowtile - cls * kp.mtiles);
      const int ntaps = __builtin_amdgcn_readfirstlane(p.cls[cls].ntaps);
      if (ntaps == 0) {
        ++L_seq;
        continue;
      }
      unsigned inv[4];
      int ah[4], aw[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int h = x >> 1, i = x & 1, j = 2 * wave + i;
        const int r = (j >> 3) * TM + h * 64 + 8 * (j & 7) + prow;  // tile row
        const int m = mt * BM + r;
        const bool real = TM == 128 || !(h == 1 && 8 * (j & 7) >= TM - 64);
        inv[x] = 0xffffffffu;
        ah[x] = aw[x] = 0;
        a_off[x] = 0;
        if (real && m < Msub) {
          int n, ii, jj;
          decode(m, n, ii, jj);
          ah[x] = ii * p.IS;
          aw[x] = jj * p.IS;
          inv[x] = 0;
          const int chunk = pch ^ ((prow >> 1) | (i << 2));
          a_off[x] = (unsigned)((n * p.Hin + ah[x]) * p.Win + aw[x]) * (unsigned)(p.pix_stride * EB) + (unsigned)(chunk * 16);
        }
      }
      const bool newcls = cls != L_cls;
      // this lane's k-tiles lane and lane + 64 of the class: (tap, chunk) in the launch's k order
      int tp0 = 0, ch0 = 0, tp1 = 0, ch1 = 0;
      if (newcls) {
        const int k0 = lane, k1 = lane + 64;
        if (kp.korder == 0) {
          tp0 = k0 / kcpt; ch0 = k0 - tp0 * kcpt;
          tp1 = k1 / kcpt; ch1 = k1 - tp1 * kcpt;
        } else {
          ch0 = k0 / ntaps; tp0 = k0 - ch0 * ntaps;
          ch1 = k1 / ntaps; tp1 = k1 - ch1 * ntaps;
        }
        v_tabA0 = v_tabA1 = v_tabB0 = v_tabB1 = 0;
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < ntaps) {
          const Tap tp = p.cls[cls].taps[t];
          if (newcls) {
            const int dA = (tp.dh * p.Win + tp.dw) * p.pix_stride * EB;
            const int dB = tp.wtap * p.Ck * EB;
            v_tabA0 = tp0 == t ? dA + ch0 * 128 : v_tabA0;
            v_tabA1 = tp1 == t ? dA + ch1 * 128 : v_tabA1;
            v_tabB0 = tp0 == t ? ((dB + ch0 * 128) | (31 - t)) : v_tabB0;
            v_tabB1 = tp1 == t ? ((dB + ch1 * 128) | (31 - t)) : v_tabB1;
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const bool ok = ((unsigned)(ah[x] + tp.dh) < (unsigned)p.Hin) && ((unsigned)(aw[x] + tp.dw) < (unsigned)p.Win);
            inv[x] |= ok ? 0u : (1u << t);
          }
        }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) a_inv[x] = inv[x];
      L_cls = cls;
      L_nt = 0;
      L_kt = 0;
      L_kend = ntaps * kcpt;
      L_ntile();
      L_tile();
      return;
    }
  };
  // A piece x = h*2 + i: the row's offset, or an out-of-range one (zeros) when the tap leaves the image
  auto L_pieceA = [&](auto xc) __attribute__((always_inline)) {
    constexpr int x = decltype(xc)::value, h = x >> 1, i = x & 1;
    // (readfirstlane: the cursor state is wave-uniform by construction; this keeps it in SGPRs for the asm operands even
    // where hipcc's divergence analysis has given up on it — free when the value already is an SGPR)
    const unsigned t = a_inv[x] << __builtin_amdgcn_readfirstlane(S_sh);  // bit 31 = "invalid for this tap"
    const unsigned off = (t & 0x80000000u) | (a_off[x] + (unsigned)__builtin_amdgcn_readfirstlane(S_dA));
    if (TM != 128 && h == 1 && (i == 0 ? a_pad0 : a_pad1))
      blds16(srdA, off, lds0 + SINK);
    else
      blds16z<i * A_STEP_I + h * A_STEP_H>(srdA, off, __builtin_amdgcn_readfirstlane(S_baseA));
  };
  auto L_pieceB = [&](auto xc) __attribute__((always_inline)) {
    constexpr int x = decltype(xc)::value;
    if constexpr (NPH == 4) {
      constexpr int h = x >> 1, i = x & 1;
      blds16o<i * B_STEP_I + h * B_STEP_H>(srdB, b_off[i], __builtin_amdgcn_readfirstlane(S_dB + (unsigned)(h * 32) * w_row),
                                             __builtin_amdgcn_readfirstlane(S_baseB));
    } else {
      blds16o<x * B_STEP_H>(srdB, b_off[0], __builtin_amdgcn_readfirstlane(S_dB + (unsigned)(x * 64) * w_row),
                            __builtin_amdgcn_readfirstlane(S_baseB));
    }
  };
  auto L_advance = [&]() __attribute__((always_inline)) {  // after the last group of a k-tile
    L_stage = L_stage + 1 == NSTG ? 0 : L_stage + 1;
    if (++L_kt == L_kend) {
      L_kt = 0;
      if (++L_nt == kp.ntpg) {
        ++L_seq;
        L_setup();  // (ends with L_tile)
        return;
      }
      L_ntile();
    }
    L_tile();
  };
  constexpr std::integral_constant<int, 2> I2{};
  constexpr std::integral_constant<int, 3> I3{};
  // issue group `j` (compile-time) of the loader's current k-tile (the caller moves the cursor after the last group)
  auto L_group = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    if constexpr (NPH == 4) {
      if constexpr (j == 0) { L_pieceA(I0); L_pieceA(I1); }
      if constexpr (j == 1) { L_pieceB(I0); L_pieceB(I1); }
      if constexpr (j == 2) { L_pieceB(I2); L_pieceB(I3); }
      if constexpr (j == 3) { L_pieceA(I2); L_pieceA(I3); }
    } else {
      if constexpr (j == 0) { L_pieceA(I0); L_pieceA(I1); L_pieceB(I0); }
      if constexpr (j == 1) { L_pieceB(I1); L_pieceA(I2); L_pieceA(I3); }
    }
  };

  // ---- fragment addresses -----------------------------------------------------------------------------------------
  const int px = lane & 15, q = lane >> 4;
  const int rsw = (px >> 1) & 7;
  int fa[2], fb[2];  // byte offsets inside a stage of this lane's fragment of tile 0, k-substep 0 / 1
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((q + 4 * ks) ^ rsw) << 4;
    fa[ks] = (wr * TM + px) * 128 + ch;
    fb[ks] = A_BYTES + (wn * TN + px) * 128 + ch;
  }

  // ---- prologue -----------------------------------------------------------------------------------------------------
  if constexpr (STATS != 0) {
    for (int i = tid; i < 2 * kp.ntpg * BN * 2; i += 512) stat_acc[i] = 0.f;
  }
  L_setup();
  // the loader runs NSTG - 1 k-tiles ahead of the MFMAs (FAT, BN 256: one k-tile + the first group of the next)
  bool all_pre = true;
#pragma unroll
  for (int pre = 0; pre < NSTG - 1; ++pre) {
    all_pre = all_pre && L_valid;
    if (L_valid) {
      L_group(I0);
      L_group(I1);
      if constexpr (NPH == 4) {
        L_group(I2);
        L_group(I3);
      }
      L_advance();
    }
  }
  if constexpr (FAT && BN == 256) {
    all_pre = all_pre && L_valid;
    if (L_valid) {  // A rows of half 0 + all weight rows of k-tile 1
      L_group(I0);
      L_group(I1);
      L_group(I2);
    }
  }
  // everything the first phase reads has landed, for every wave
  if (all_pre) wait_vm8<FAT ? WAITN : (D - 2) * GL>(); else wait_vm8<0>();
  MI355_LDS_BARRIER();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs one barrier behind

  bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* addend = reinterpret_cast<const bf16_t*>(p.addend);
#ifdef MI355_ITEMSTAMP
  int item_n = 0;
#endif
  int stage = 0;
  bool after_epi = false;  // the epilogue's NST stores are younger than the groups the next k-tile's waits retire

  for (int seq = 0;; ++seq) {
    const int item = bx + seq * G;
    if (item >= kp.items) break;
    const int rowtile = item / kp.ngroups;
    const int grp = item - rowtile * kp.ngroups;
    const int ci = rowtile / kp.mtiles;
    const int mtile = rowtile - ci * kp.mtiles;
    const TapClass& cls = p.cls[ci];
    const int nk = cls.ntaps * kcpt;
    const int m0 = mtile * BM;
    for (int nti = 0; nti < kp.ntpg; ++nti) {
      const int n0 = (grp * kp.ntpg + nti) * BN;
      f32x4 acc[MT][NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

      // One k-tile.  `mode` picks the counted waits: 0 steady state; 1 first k-tile after an epilogue (its NST stores are
      // younger than the groups those waits retire); 2 the loader has run out (nothing is issued, every wait drains).
      // (ONE copy of the body, the mode a wave-uniform runtime value: three copies of a loop that keeps 128 accumulator
      // registers live across them send hipcc's register allocator into hundreds of spills.)
      //   slim, BN 256: 4 phases x 16 MFMAs; phase p issues group p of the NEXT k-tile; the cursor moves after phase 4
      //   slim, BN 128: 2 phases x 16 MFMAs, 3 stages: phase p issues group p of the k-tile two ahead
      //   fat,  BN 256: 2 phases x 32 MFMAs; phase 1 issues the A rows of half 1 of the next k-tile, moves the cursor,
      //                 phase 2 the A rows of half 0 + all weight rows of the k-tile after that (each region is rewritten
      //                 ONE phase after its last read: legal because every wave's lgkmcnt(0) sits before the first
      //                 barrier of the reading phase); 1.5 k-tiles of lookahead in a 2-stage ring
      //   fat,  BN 128: 1 phase x 32 MFMAs, 3 stages: issues the whole k-tile two ahead
      auto ktile = [&](const int mode) __attribute__((always_inline)) {
        auto wait_new = [&]() __attribute__((always_inline)) {  // retires groups issued after the last epilogue
          if (mode == 2) wait_vm8<0>(); else wait_vm8<WAITN>();
        };
        auto wait_old = [&]() __attribute__((always_inline)) {  // retires groups issued before it
          if (mode == 0) wait_vm8<WAITN>(); else if (mode == 1) wait_vm8<WAITN + NST>(); else wait_vm8<0>();
        };
        const char* sb = smem + stage * STAGE;
        const char* pa0 = sb + fa[0];
        const char* pa1 = sb + fa[1];
        const char* pb0 = sb + fb[0];
        const char* pb1 = sb + fb[1];
        constexpr int AFN = (FAT && BN == 128) ? MT : MT0;
        bf16x8 af[AFN][2], bf0[2][2], bf1[2][2];
        auto rdA = [&](auto hc) __attribute__((always_inline)) {
          constexpr int half = decltype(hc)::value;
          constexpr int base = (FAT && BN == 128) ? half * MT0 : 0;  // fat BN 128 keeps both halves in registers
#pragma unroll
          for (int mt = 0; mt < (half == 0 ? MT0 : MT1); ++mt) {
            af[base + mt][0] = *reinterpret_cast<const bf16x8*>(pa0 + (half * MT0 + mt) * 2048);
            af[base + mt][1] = *reinterpret_cast<const bf16x8*>(pa1 + (half * MT0 + mt) * 2048);
          }
        };
        auto rdB = [&](bf16x8 (&bf)[2][2], auto hc) __attribute__((always_inline)) {
          constexpr int half = decltype(hc)::value;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            bf[nt][0] = *reinterpret_cast<const bf16x8*>(pb0 + (half * 2 + nt) * 2048);
            bf[nt][1] = *reinterpret_cast<const bf16x8*>(pb1 + (half * 2 + nt) * 2048);
          }
        };
        auto mma = [&](auto mc, bf16x8 (&bf)[2][2], auto nc) __attribute__((always_inline)) {
          constexpr int mhalf = decltype(mc)::value, nhalf = decltype(nc)::value;
          constexpr int base = (FAT && BN == 128) ? mhalf * MT0 : 0;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < (mhalf == 0 ? MT0 : MT1); ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                if constexpr (EB == 2) {
                  acc[mhalf * MT0 + mt][nhalf * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      bf[nt][ks], af[base + mt][ks], acc[mhalf * MT0 + mt][nhalf * 2 + nt], 0, 0, 0);
                } else {
                  // the 2 x 16 bytes a lane read are the operands of four k-steps of 32 (the SAME bytes of the weight and the pixel row on both sides, so
                  // the assignment of k values to lanes need not be known: k is a summation index) = ONE k-step of the 16x16x128 f8f6f4 form, which
                  // runs at twice the rate of four 16x16x32 fp8 instructions (tools/micro/mfma_rate.hip: 5.0 against 2.1 PFLOP/s).
                  if (ks == 0) {  // (compile-time under the unroll: the instruction consumes both halves of the k-tile)
                    typedef int i32x8 __attribute__((ext_vector_type(8)));
                    const i32x4 w0 = __builtin_bit_cast(i32x4, bf[nt][0]), w1 = __builtin_bit_cast(i32x4, bf[nt][1]);
                    const i32x4 x0 = __builtin_bit_cast(i32x4, af[base + mt][0]), x1 = __builtin_bit_cast(i32x4, af[base + mt][1]);
                    const i32x8 wv = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    const i32x8 xv = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    // (as text: the builtin of the block-scaled form takes its accumulators in VGPRs beside scale registers and the kernel spills 64-193
                    //  dwords; the non-scaled form with cbsz = blgp = 0 is e4m3 x e4m3 and needs no scale operands)
                    f32x4 c = acc[mhalf * MT0 + mt][nhalf * 2 + nt];
                    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(wv), "v"(xv));
                    acc[mhalf * MT0 + mt][nhalf * 2 + nt] = c;
                  }
                }
        };
        // one phase: [fragment reads][issue][counted wait] lgkmcnt(0) barrier [MFMAs] barrier
#ifdef MI355_STAMP8
#define STAMP8_NEXT() ++stamp_n
#else
#define STAMP8_NEXT()
#endif
#define MI355_PHASE_MID()                                  \
  STAMP8(1);                                               \
  __builtin_amdgcn_sched_barrier(0);                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_s_barrier();                            \
  __builtin_amdgcn_sched_barrier(0);                       \
  STAMP8(2);                                               \
  __builtin_amdgcn_s_setprio(1);
#define MI355_PHASE_END()                                  \
  __builtin_amdgcn_s_setprio(0);                           \
  STAMP8(3);                                               \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_barrier();                            \
  asm volatile("" ::: "memory");                           \
  __builtin_amdgcn_sched_barrier(0);                       \
  STAMP8_NEXT();
        if constexpr (!FAT && NPH == 4) {
          // phase 1: quadrant (m half 0, n half 0)
          STAMP8(0);
          rdB(bf0, I0);
          rdA(I0);
          if (mode != 2) L_group(I0);
          wait_old();  // group 2 of this k-tile (weight rows of n half 1), read in phase 2
          MI355_PHASE_MID();
          mma(I0, bf0, I0);
          MI355_PHASE_END();
          // phase 2: (m half 0, n half 1)
          STAMP8(0);
          rdB(bf1, I1);
          if (mode != 2) L_group(I1);
          wait_old();  // group 3 of this k-tile (A rows of m half 1), read in phase 3
          MI355_PHASE_MID();
          mma(I0, bf1, I1);
          MI355_PHASE_END();
          // phase 3: (m half 1, n half 1)
          STAMP8(0);
          rdA(I1);
          if (mode != 2) L_group(I2);
          MI355_PHASE_MID();
          mma(I1, bf1, I1);
          MI355_PHASE_END();
          // phase 4: (m half 1, n half 0): fragments already in registers; the cursor moves under the MFMAs
          STAMP8(0);
          if (mode != 2) L_group(I3);
          wait_new();  // groups 0, 1 of the next k-tile (issued in phases 1, 2 of this one), read in its phase 1
          MI355_PHASE_MID();
          mma(I1, bf0, I0);
          if (mode != 2) L_advance();
          MI355_PHASE_END();
        } else if constexpr (!FAT) {
          // phase 1: m half 0 x both 16-channel tiles
          STAMP8(0);
          rdB(bf0, I0);
          rdA(I0);
          if (mode != 2) L_group(I0);
          MI355_PHASE_MID();
          mma(I0, bf0, I0);
          MI355_PHASE_END();
          // phase 2: m half 1
          STAMP8(0);
          rdA(I1);
          if (mode != 2) L_group(I1);
          wait_old();  // both groups of the next k-tile (issued one k-tile ago)
          MI355_PHASE_MID();
          mma(I1, bf0, I0);
          if (mode != 2) L_advance();
          MI355_PHASE_END();
        } else if constexpr (BN == 256) {
          // phase 1: m half 0 x all 4 channel tiles
          STAMP8(0);
          rdB(bf0, I0);
          rdB(bf1, I1);
          rdA(I0);
          if (mode != 2) {
            L_group(I3);  // A rows of half 1 of the loader's k-tile (the next one) ...
            L_advance();  // ... which completes it
          }
          wait_old();  // A rows of half 1 of THIS k-tile, read in phase 2
          MI355_PHASE_MID();
          mma(I0, bf0, I0);
          mma(I0, bf1, I1);
          MI355_PHASE_END();
          // phase 2: m half 1
          STAMP8(0);
          rdA(I1);
          if (mode != 2 && L_valid) {
            L_group(I0);
            L_group(I1);
            L_group(I2);
            wait_old();  // A rows of half 0 + weight rows of the next k-tile, read in its phase 1
          } else {
            wait_vm8<0>();
          }
          MI355_PHASE_MID();
          mma(I1, bf1, I1);
          mma(I1, bf0, I0);
          MI355_PHASE_END();
        } else {
          STAMP8(0);
          rdB(bf0, I0);
          rdA(I0);
          rdA(I1);
          if (mode != 2) {
            L_group(I0);
            L_group(I1);
          }
          wait_old();  // the next k-tile (issued one k-tile ago)
          MI355_PHASE_MID();
          mma(I0, bf0, I0);
          mma(I1, bf0, I0);
          if (mode != 2) L_advance();
          MI355_PHASE_END();
        }
        stage = stage + 1 == NSTG ? 0 : stage + 1;
      };
      ITEMSTAMP(0);
      for (int kt = 0; kt < nk; ++kt) {
        ktile(!L_valid ? 2 : (after_epi ? 1 : 0));
        after_epi = false;
      }
      ITEMSTAMP(1);

      // (the e4m3 path issues its matrix instructions as text: the compiler does not know their result latency, so the wait states in front of the
      //  first read of an accumulator are spelled out)
      if constexpr (EB == 1) {
        __builtin_amdgcn_sched_barrier(0);  // (nothing that reads an accumulator may be scheduled above the wait states)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue: both wave rows in the same barrier interval ---------------------------------------------------
      if (wr == 0) __builtin_amdgcn_s_barrier();
      {
        const int cb = (q & 1) * 16 + (q >> 1) * 8;  // this lane's 8 channels inside a pair of 16-channel tiles
        const bool odd = (q & 1) != 0;
#pragma unroll
        for (int ntp = 0; ntp < NT / 2; ++ntp) {
          const int c0 = n0 + wn * TN + ntp * 32 + cb;
          float s1[8], s2[8], bmu[8], bis[8];
          if constexpr (STATS == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(p.bn_mean + c0 + 4 * h);
              const f32x4 b = *reinterpret_cast<const f32x4*>(p.bn_invstd + c0 + 4 * h);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                bmu[4 * h + e] = a[e];
                bis[4 * h + e] = b[e];
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
          constexpr int MB = 4;  // pixel tiles per batch: all loads of a batch first
#pragma unroll
          for (int mb = 0; mb < MT; mb += MB) {
            int pixs[MB];
            uint4 araw[MB], yraw[MB];
            unsigned abits[MB], ybits[MB];
#pragma unroll
            for (int u = 0; u < MB; ++u) {
              const int mt = mb + u;
              if (mt >= MT) continue;
              const int m = m0 + wr * TM + mt * 16 + px;
              int pix = -1;
              if (m < Msub) {
                if (p.OS == 1) {
                  pix = m;
                } else {
                  int n, ii, jj;
                  decode(m, n, ii, jj);
                  pix = (n * p.Hout + ii * p.OS + cls.ph) * p.Wout + jj * p.OS + cls.pw;
                }
              }
              pixs[u] = pix;
              const size_t o = (size_t)(pix < 0 ? 0 : pix) * p.Ncols + c0;
              if (addend) {
                araw[u] = *reinterpret_cast<const uint4*>(addend + o);
                abits[u] = p.addend_bits ? (unsigned)p.addend_bits[o / 8] : 0xffu;
              }
              if constexpr (STATS == 2) {
                yraw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bn_y) + o);
                ybits[u] = (unsigned)p.bn_bits[o / 8];
              }
            }
#pragma unroll
            for (int u = 0; u < MB; ++u) {
              const int mt = mb + u;
              if (mt >= MT) continue;
              f32x4 a = acc[mt][2 * ntp], b = acc[mt][2 * ntp + 1];
              if constexpr (EB == 1) {  // undo the operands' quantisation scales
                a *= oscale;
                b *= oscale;
              }
              float v[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float send = odd ? a[e] : b[e];
                const float recv = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), 0x401F));  // lane ^ 16
                v[e] = odd ? recv : a[e];
                v[4 + e] = odd ? b[e] : recv;
              }
              const int pix = pixs[u];
              if (addend) {
                float ad[8];
                Vec16<bf16_t>::unpack(araw[u], ad);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (pix >= 0 && ((abits[u] >> e) & 1u)) ? ad[e] : 0.f;
              }
              const size_t o = (size_t)(pix < 0 ? 0 : pix) * p.Ncols + c0;
              bf16_t* dst = pix < 0 ? reinterpret_cast<bf16_t*>(g8_trash + tid * 16) : out + o;
              Vec16<bf16_t>::store(dst, v);
              if constexpr (STATS == 1) {
                if (pix >= 0) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    const float xr = (float)(bf16_t)v[e];
                    s1[e] += xr;
                    s2[e] += xr * xr;
                  }
                }
              }
              if constexpr (STATS == 2) {
                if (pix >= 0) {
                  float yv[8];
                  Vec16<bf16_t>::unpack(yraw[u], yv);
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    const float dz = (ybits[u] >> e) & 1u ? (float)(bf16_t)v[e] : 0.f;
                    s1[e] += dz;
                    s2[e] += dz * ((yv[e] - bmu[e]) * bis[e]);
                  }
                }
              }
            }
          }
          if constexpr (STATS != 0) {
            // sum over the 16 pixels of a lane row (fixed order); lane 15 of the row owns the 8 channel slots of this wave
            float* slot = stat_acc + ((wr * kp.ntpg + nti) * BN + wn * TN + ntp * 32 + cb) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = row_sum16(s1[e]);
              const float b = row_sum16(s2[e]);
              if (px == 15) {
                slot[2 * e] += a;
                slot[2 * e + 1] += b;
              }
            }
          }
        }
      }
      after_epi = true;
      asm volatile("" ::: "memory");
      ITEMSTAMP(2);
#ifdef MI355_ITEMSTAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // profiling build: how long the stores take to retire
      ITEMSTAMP(3);
      ++item_n;
#endif
      if (wr == 1) __builtin_amdgcn_s_barrier();
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with wave row 1's entry barrier
  if constexpr (STATS != 0) {
    MI355_LDS_BARRIER();
    const int chan = kp.ntpg * BN;
    const int grp = bx % kp.ngroups;
    float* row = p.stat_partial + (size_t)(bx / kp.ngroups) * 2 * p.Ncols + grp * chan;
    for (int c = tid; c < chan; c += 512) {
      const float a = stat_acc[c * 2] + stat_acc[(chan + c) * 2];
      const float b = stat_acc[c * 2 + 1] + stat_acc[(chan + c) * 2 + 1];
      row[c] = a;
      row[p.Ncols + c] = b;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MI355_STAMP8
  if (blockIdx.x == 0 && lane == 0 && wn == 0)
    for (int i = 0; i < 1024; ++i)
      g8_stamps[wr * 1024 + i] = reinterpret_cast<unsigned long long*>(smem + STAMP_OFF)[wr * 1024 + i];
#endif
}

unsigned magic32(unsigned d) { return (unsigned)((1ull << 32) / d + 1); }

template <int BM, int BN, int FAT, int EB = 2>
int launch8_t(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows, int korder, float oscale = 1.f) {
  static_assert(EB == 2 || EB == 1, "EB");
  constexpr int NSTG = BN == 256 ? 2 : 3;
  const int MAX_WG = device_cus();  // one persistent workgroup per CU
  Igemm8KArgs k;
  k.a = a;
  const int Msub = a.N * a.Hsub * a.Wsub;
  k.mtiles = cdiv(Msub, BM);
  const int ny = a.Ncols / BN;
  const int R = nclass * k.mtiles;
  int ng = ny;
  for (int d = 1; d <= ny; ++d)
    if (ny % d == 0 && R * d >= MAX_WG) {
      ng = d;
      break;
    }
  // one n-tile per item + XCD-aware item order where there are >= 2 n-tiles:
  k.xcd = 0;
  // measured per layer (profiles/r03f): neutral or +2...+12 us on the forward launches although their L2-side traffic drops
  // (layer-3 conv3 270 -> 218 MB), -51 us on the stride-2 downsample dgrad of layer 4 (4 tap classes, 3 of them empty) and
  // -6 us on layer 3's: on by default for multi-class launches only.
  if (nclass > 1 && ny >= 2 && R * ny >= MAX_WG) {
    ng = ny;
    k.xcd = 1;
  }
  // a workgroup keeps the statistics of its n-tile group in LDS (<= 512 channels): where the rows alone fill the chip (ng = 1) and the launch has more
  // columns than that (conv3 of layers 3 / 4 in the e4m3 step: 1024 / 2048), split the columns into groups rather than dropping the epilogue statistics —
  // the pixel tile is read once per group (25 MB more at layer 3, batch 512) instead of a stand-alone pass over the output (205 MB + a launch)
  if (a.stat_partial != nullptr && (ny / ng) * BN > 512)
    for (int d = ng; d <= ny; ++d)
      if (ny % d == 0 && (ny / d) * BN <= 512) {
        ng = d;
        break;
      }
  k.ngroups = ng;
  k.ntpg = ny / ng;
  k.items = R * ng;
  k.HW = a.Hsub * a.Wsub;
  k.magW = magic32((unsigned)a.Wsub);
  k.magHW = magic32((unsigned)k.HW);
  k.korder = korder;
  k.oscale = oscale;
  k.sc_in = a.q_scale_in;
  k.sc_wt = a.q_scale_wt;
  const size_t bytes_in = (size_t)a.N * a.Hin * a.Win * a.pix_stride * EB;
  const size_t bytes_wt = (size_t)a.Ncols * a.wtaps * a.Ck * EB;
  MI355_ARG(bytes_in < 0x80000000ull && bytes_wt < 0x80000000ull, "igemm8: tensor exceeds the 2 GiB buffer-offset range");
  k.bytes_in = (unsigned)bytes_in;
  k.bytes_wt = (unsigned)bytes_wt;
#ifdef MI355_PROBES
  // timing-only probes (profiling build, `make probes`): a descriptor with zero records drops every load through it
  if (const int dbg = probe_env("MI355_IGEMM8_DBG")) {
    if (dbg & 1) k.bytes_in = 0;
    if (dbg & 2) k.bytes_wt = 0;
  }
#endif
  int grid = k.items < MAX_WG ? k.items : MAX_WG;
  grid -= grid % ng;  // items is a multiple of ng; a workgroup then keeps one n-tile group (statistics rows, weight reuse per XCD)
  if (grid == 0) grid = ng;
  const int chan = k.ntpg * BN;
  size_t lds = (size_t)NSTG * (BM + BN) * 128 + 1024;
#ifdef MI355_STAMP8
  lds += 2 * 1024 * 8;
  MI355_ARG(a.stat_partial == nullptr && lds <= 160 * 1024, "igemm8: the stamp build runs plain launches of the 2-stage tiles only");
#endif
  const bool stats = a.stat_partial != nullptr && chan <= 512 && lds + (size_t)2 * chan * 2 * sizeof(float) <= 160 * 1024;
  if (stat_rows) *stat_rows = stats ? grid / ng : 0;
  if (stats) {
    lds += (size_t)2 * chan * 2 * sizeof(float);
    if (a.bn_y) {
      lds_opt_in8((const void*)igemm8_kernel<BM, BN, 2, FAT, EB>, lds);
      hipLaunchKernelGGL((igemm8_kernel<BM, BN, 2, FAT, EB>), dim3(grid), dim3(512), lds, stream, k);
    } else {
      lds_opt_in8((const void*)igemm8_kernel<BM, BN, 1, FAT, EB>, lds);
      hipLaunchKernelGGL((igemm8_kernel<BM, BN, 1, FAT, EB>), dim3(grid), dim3(512), lds, stream, k);
    }
  } else {
    lds_opt_in8((const void*)igemm8_kernel<BM, BN, 0, FAT, EB>, lds);
    hipLaunchKernelGGL((igemm8_kernel<BM, BN, 0, FAT, EB>), dim3(grid), dim3(512), lds, stream, k);
  }
  MI355_LAUNCH_CHECK();
  note_kernel("igemm8<%d,%d%s%s>", BM, BN, FAT ? ",fat" : "", EB == 1 ? ",e4m3" : "");
  return 0;
}

}  // namespace

// true when the 8-wave kernel can run this launch at all (bf16 only)
bool igemm8_legal(const IgemmArgs& a, int nclass, int bn) {
  if (a.pair_delta != 0 || a.Ck % 64 != 0 || a.Ncols % bn != 0 || a.Wsub < 2 || a.Hsub < 1) return false;
  for (int ci = 0; ci < nclass; ++ci)
    if (a.cls[ci].ntaps * (a.Ck / 64) > 128) return false;  // the per-class k-tile tables hold 128 entries
  const unsigned long long Msub = (unsigned long long)a.N * a.Hsub * a.Wsub;
  const unsigned long long hw = (unsigned long long)a.Hsub * a.Wsub;
  if (Msub * hw >= (1ull << 32) || Msub + 256 >= (1ull << 31)) return false;
  return true;
}

// fp8 (OCP e4m3) operands, bf16 output = conv(xq, wq) * oscale: the same kernel with EB = 1 (k-tiles of 128 channels)
// true when the fp8 form of the 8-wave kernel can run this launch (k-tiles of 128 channels)
bool igemm8_fp8_legal(const IgemmArgs& a, int nclass, int bn) {
  if (a.pair_delta != 0 || a.Ck % 128 != 0 || a.Ncols % bn != 0 || a.Wsub < 2 || a.Hsub < 1) return false;
  for (int ci = 0; ci < nclass; ++ci)
    if (a.cls[ci].ntaps * (a.Ck / 128) > 128) return false;
  const unsigned long long Msub = (unsigned long long)a.N * a.Hsub * a.Wsub;
  return Msub * a.Hsub * a.Wsub < (1ull << 32) && Msub + 256 < (1ull << 31);
}

int launch_igemm8_fp8(const IgemmArgs& a, int nclass, int bm, int bn, int korder, float oscale, hipStream_t stream, int* stat_rows) {
  MI355_ARG(a.pair_delta == 0 && a.Ck % 128 == 0 && a.Ncols % bn == 0 && a.Wsub >= 2, "igemm8 fp8: Ck=%d (multiple of 128), Ncols=%d (multiple of %d), output rows of %d pixels (at least 2)", a.Ck, a.Ncols, bn, a.Wsub);
  for (int ci = 0; ci < nclass; ++ci) MI355_ARG(a.cls[ci].ntaps * (a.Ck / 128) <= 128, "igemm8 fp8: more than 128 k-tiles per class");
  const unsigned long long Msub = (unsigned long long)a.N * a.Hsub * a.Wsub;
  MI355_ARG(Msub * a.Hsub * a.Wsub < (1ull << 32), "igemm8 fp8: problem too large for the 32-bit index arithmetic");
  if (bm == 224 && bn == 256) return launch8_t<224, 256, 0, 1>(a, nclass, stream, stat_rows, korder, oscale);
  if (bm == 256 && bn == 256) return launch8_t<256, 256, 0, 1>(a, nclass, stream, stat_rows, korder, oscale);
  if (bm == 256 && bn == 128) return launch8_t<256, 128, 1, 1>(a, nclass, stream, stat_rows, korder, oscale);
  set_error("igemm8 fp8: no %dx%d tile", bm, bn);
  return MI355_E_ARG;
}

int launch_igemm8(const IgemmArgs& a, int nclass, int bm, int bn, int korder, int fat, hipStream_t stream, int* stat_rows) {
  MI355_ARG(igemm8_legal(a, nclass, bn), "igemm8: unsupported geometry");
  if (bm == 256 && bn == 256) return fat ? launch8_t<256, 256, 1>(a, nclass, stream, stat_rows, korder) : launch8_t<256, 256, 0>(a, nclass, stream, stat_rows, korder);
  if (bm == 224 && bn == 256) return fat ? launch8_t<224, 256, 1>(a, nclass, stream, stat_rows, korder) : launch8_t<224, 256, 0>(a, nclass, stream, stat_rows, korder);
  if (bm == 256 && bn == 128) return fat ? launch8_t<256, 128, 1>(a, nclass, stream, stat_rows, korder) : launch8_t<256, 128, 0>(a, nclass, stream, stat_rows, korder);
  if (bm == 224 && bn == 128) return fat ? launch8_t<224, 128, 1>(a, nclass, stream, stat_rows, korder) : launch8_t<224, 128, 0>(a, nclass, stream, stat_rows, korder);
  set_error("igemm8: no %dx%d tile", bm, bn);
  return MI355_E_ARG;
}

}  // namespace mi355

#ifdef MI355_ITEMSTAMP
extern "C" int mi355_debug_items8(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mi355::g8_items), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
extern "C" int mi355_debug_items8_clear(void) {
  static unsigned long long z[512 * 16 * 4];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi355::g8_items), z, sizeof(z), 0, hipMemcpyHostToDevice);
}
#endif

#ifdef MI355_STAMP8
extern "C" int mi355_debug_stamps8(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mi355::g8_stamps), (size_t)n * sizeof(unsigned long long), 0,
                                  hipMemcpyDeviceToHost);
}
#endif
