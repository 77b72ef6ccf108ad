// conv_igemm.hip — implicit-GEMM convolution forward / data-gradient for gfx950 (MI355X).
//
// Replaces the cuDNN conv fwd / dgrad kernels that run beneath `model(data)` and `loss.backward()` in the
// reference (call form sota_imagenet/callbacks.py:316-317; model built at train.py:64).
//
// One kernel serves forward, dgrad (stride 1 and, through four output-parity classes, stride 2), the 7x7 stem
// (4 row-pair taps over a zero-padded NHWC4 image) and the FC layer: see IgemmArgs in common.h.
//
//   tile        BM pixels x BN channels; K advances in 128-byte slabs of the tap's channel run.  Instantiations (see the
//               kernel's own comment for why each exists): 128 x 64 and 128 x 128 (4 waves, 2-stage ring, 3 resp. 2
//               workgroups per CU — the default), 256 x 256 (4 waves, 1 per CU) and 256 x 128 (8 waves, 3-stage ring,
//               1 per CU) for the bf16 launches the rules in launch_igemm() select.
//   MFMA        waves in a WMW x 2 grid, each BM/WMW x BN/2 from 32x32 tiles: v_mfma_f32_32x32x2_f32 (exact fp32, the
//               parity path) / v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Operands are swapped (D^T = W * A^T) so a
//               lane ends up holding 4 consecutive channels of one pixel.
//   staging     direct-to-LDS buffer loads (buffer_load_dwordx4 ... offen lds): every lane supplies its own 32-bit byte
//               offset into the tensor's buffer descriptor (the A rows are gathered pixel runs; out-of-image rows get
//               an out-of-range offset and the hardware range check returns zeros), the LDS image stays lane-linear
//               ([row][128 B], no padding) and the bank-conflict fix is an XOR of the 16-byte chunk index with
//               (row>>1)&7 applied to the SOURCE offset and again on the fragment ds_read_b128.  The loads of a later
//               slab are issued between the four MFMA groups of the current one, so the matrix pipe keeps running
//               while the loader computes offsets.
//   pipeline    persistent workgroups walk (row-tile, n-tile, k-slab) as ONE stream through an NSTG-stage LDS ring:
//               the loads of the next NSTG-1 slabs — also across tile boundaries, i.e. under the previous tile's
//               epilogue — are in flight while the current slab feeds the MFMAs; counted `s_waitcnt vmcnt(N)` + raw
//               s_barrier (never __syncthreads, whose fence would drain the LDS-DMA queue).
//   epilogue    each wave stages its 32 x BN/2 fp32 sub-tile through (XOR-swizzled) LDS and writes 16 bytes per lane,
//               64..256 contiguous bytes per pixel; the optional addend (residual gradient, under a ReLU bit mask) is
//               read the same way; optionally BN statistics of the output / the BN-backward sums of the layer the
//               output is the activation gradient of (per-workgroup LDS accumulators, one flush per workgroup).
//   fp32 only   stream-K over the partial last round of the 128 x 128 launches (see IgemmKArgs).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>

#include "common.h"
#include "lds_dma.h"
#include "vec.h"

namespace mi355 {

namespace {

constexpr int BKB = 128;  // bytes of K per slab (= one LDS row)

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];
// -DMI355_STAMP: cycle stamps of workgroup 0 inside the main loop, read back by tools/stamp_conv.py (profiling builds only)
#ifdef MI355_STAMP
__device__ unsigned long long g_stamps[8 * 4096];  // [wave 0..3 of workgroup 0][k-step][8 stamps]
#define STAMP(slot)                                                                                   \
  do {                                                                                                \
    if (blockIdx.x == 0 && lane == 0 && wave < 4 && stamp_k < 1024) g_stamps[(wave * 1024 + stamp_k) * 8 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define STAMP(slot)
#endif
__device__ __attribute__((aligned(256))) unsigned char g_trash[512 * 16];
// -DMI355_PROBES (make PROBES=1 -> lib/variant_probes.so): the epilogue timing probes of tools/probe8.py (MI355_IGEMM_DBG); they cost
// registers in the hottest epilogue, so the product build compiles them out
#ifdef MI355_PROBES
#define MI355_PROBE(bit) ((kp.dbg & (bit)) != 0)
#else
#define MI355_PROBE(bit) false
#endif

struct IgemmKArgs {
  IgemmArgs a;
  int mtiles, ny, nclass, ngroups, ntpg, items;
  unsigned bytes_in, bytes_wt;
  // stream-K over the last, partial round (sk_tail > 0; needs nclass == 1 and ntpg == 1): per n-tile group the row tiles
  // [sk_rf * Gg, sk_rf * Gg + sk_tail) are not handed out whole — their sk_tail * nk k-steps are cut into Gg = G/ngroups
  // equal ranges, one per workgroup of the group.  The workgroup that starts a tile (k = 0) owns it: it adds the fp32
  // partial tiles the following workgroups leave in sk_partial (one 16-byte-per-lane slot per workgroup, published
  // through sk_flags with the launch's epoch) in workgroup order, then runs the normal epilogue.
  int sk_rf, sk_tail, sk_nk, sk_gshift;  // sk_gshift = log2(G / ngroups)
  int sk_tshift;                         // log2 of the workgroups per group that share the tail (<= G / ngroups: a tile
                                         // is cut into at most ~4 parts, each hand-off costs its owner ~10 us)
  unsigned sk_epoch;
  float* sk_partial;
  unsigned* sk_flags;       // [G] epoch flags, then the error word at index IGEMM_SK_ERR_WORD (read by the executor)
  unsigned sk_spin_limit;   // polls before an owner gives up on a hand-off and raises the error word
  int sk_mute;              // debug (MI355_SK_DEBUG=mute): contributors never publish, so every owner times out
  int dbg;                  // timing probes (MI355_IGEMM_DBG): 1 = skip the epilogue, 2 = epilogue stores go to the trash page,
                            // 4 = the addend is read from one cached zero page (no HBM latency in the epilogue)
  int xcd;                  // 1: workgroup b takes the item stream of virtual workgroup (b % 8) * (G / 8) + b / 8, so the workgroups
                            // resident on ONE XCD (hardware round-robin: b % 8) walk CONSECUTIVE items — with one n-tile per item
                            // the n-tiles of a row tile run side by side behind one L2 (launch_t says when)
};

// BM x BN tile, WMW x 2 waves (a wave owns BM/WMW x BN/2 outputs), NSTG-stage LDS ring:
//   128 x 64 (3 workgroups per CU), 128 x 128 (2 per CU): 2 x 2 waves, 2 stages
//   256 x 256: 2 x 2 waves, 2 stages, 1 per CU; a wave owns 128 x 128 outputs, which halves the LDS bytes moved per MFMA
//   256 x 128: 4 x 2 waves, 3 stages, 1 per CU: the same per-wave work as the 128 x 128 tile, one weight slab feeds 256
//              rows (6 instead of 8 LDS-DMA pieces per wave and k-step) and the loader runs TWO slabs ahead, which is what
//              the ~700 cycles a 128 x 128 workgroup waits for its slab every k-step ask for (stamps)
// STATS: 0 none, 1 forward BN statistics, 2 BN-backward sums (see common.h)
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int WMW, int NSTG, int STATS>
__global__ __launch_bounds__(128 * WMW, (BM == 256 ? 1 : (BN == 64 && NSTG == 2 ? 3 : 2))) void igemm_kernel(const IgemmKArgs kp) {
  const IgemmArgs& p = kp.a;
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = BKB / ES;
  constexpr int NW = 2 * WMW;              // waves
  constexpr int NT = 64 * NW;              // threads
  constexpr int MI = BM / (32 * WMW);      // 32-pixel MFMA tiles per wave
  constexpr int NI = BN / 64;              // 32-channel MFMA tiles per wave
  constexpr int PA = BM / NW / 8;          // A pieces (1 KiB wave-instructions) per wave per slab
  constexpr int PB = BN / NW / 8;          // B pieces per wave per slab
  constexpr int NPC = PA + PB;
  constexpr int A_BYTES = BM * BKB;
  constexpr int STAGE = (BM + BN) * BKB;   // one ring stage
  constexpr int WN = BN / 2;               // channels per wave
  constexpr int VEC = 16 / ES;             // output elements per 16-byte store
  constexpr int CPR = WN / VEC;            // 16-byte output chunks per staged row
  constexpr int RPI = 64 / CPR;            // rows per wave-instruction in the read-back
  constexpr int NST = MI * (32 / RPI);     // global stores per thread per tile
  constexpr int SCH = WN / 4;              // 16-byte fp32 chunks per staged row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // epilogue staging: 32 x WN fp32 per wave inside the stage consumed last; when the waves do not all fit there (8 waves
  // on the 256 x 128 tile) they take turns, EPI_TURNS groups of EPI_WAVES waves
  constexpr int STG_WAVE = 32 * (BN / 2) * 4;
  constexpr int EPI_FIT = STAGE / STG_WAVE;  // staging blocks that fit into one stage
  constexpr int EPI_WAVES = NW <= EPI_FIT ? NW : (EPI_FIT >= 4 ? 4 : (EPI_FIT >= 2 ? 2 : 1));
  constexpr int EPI_TURNS = NW / EPI_WAVES;
  static_assert(NW % EPI_WAVES == 0, "epilogue turns");
  int* row_pix = reinterpret_cast<int*>(smem + NSTG * STAGE);
  // STATS: per-workgroup BN statistics of the outputs, [wave-row wm][channel of this workgroup's n-tiles][sum, sum^2]
  float* stat_acc = reinterpret_cast<float*>(smem + NSTG * STAGE + BM * sizeof(int));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const int Msub = p.N * p.Hsub * p.Wsub;
  const int kc_per_tap = p.Ck / BK;

  // ---- loader cursor: runs one slab ahead of the MFMAs ------------------------------------------------------------
  const int prow = lane >> 3;  // row within a 1 KiB piece
  const int pch = lane & 7;    // physical 16-byte chunk
  int L_seq = 0, L_item = 0, L_nt = 0, L_kt = 0, L_kend = 0, L_cls = 0, L_grp = 0;
  bool L_valid = true;
  // the seq-th unit of work of this workgroup: item (row tile x n-tile group) and k-step range [k0, k1) (k1 < 0: all).
  // With stream-K the (at most two) tail units of the workgroup are worked out once, in 32-bit scalar arithmetic
  // (sk_tail * sk_nk * G < 2^31; G / ngroups is a power of two).
  constexpr bool SKC = BM == 128 && BN == 128 && ES == 4;  // stream-K is compiled into the fp32 128 x 128 kernel only
  int sk_bg = 0, sk_n = 0, sk_item0 = 0, sk_k00 = 0, sk_k10 = 0, sk_k11 = 0;
  if constexpr (SKC) {
    if (kp.sk_tail != 0) {
      sk_bg = (int)blockIdx.x / kp.ngroups;
      const int g = (int)blockIdx.x - sk_bg * kp.ngroups;
      const unsigned TK = (unsigned)(kp.sk_tail * kp.sk_nk);
      const unsigned lo = ((unsigned)sk_bg * TK) >> kp.sk_tshift, hi = ((unsigned)(sk_bg + 1) * TK) >> kp.sk_tshift;
      if (lo < hi && sk_bg < (1 << kp.sk_tshift)) {
        const int t0 = (int)(lo / (unsigned)kp.sk_nk);
        sk_item0 = ((kp.sk_rf << kp.sk_gshift) + t0) * kp.ngroups + g;
        sk_k00 = (int)lo - t0 * kp.sk_nk;
        const int e = sk_k00 + (int)(hi - lo);
        sk_k10 = e < kp.sk_nk ? e : kp.sk_nk;
        sk_k11 = (int)hi - (t0 + 1) * kp.sk_nk;  // > 0: the range runs into the next tile, which this workgroup owns
        sk_n = sk_k11 > 0 ? 2 : 1;
      }
      sk_bg = __builtin_amdgcn_readfirstlane(sk_bg);
      sk_n = __builtin_amdgcn_readfirstlane(sk_n);
      sk_item0 = __builtin_amdgcn_readfirstlane(sk_item0);
      sk_k00 = __builtin_amdgcn_readfirstlane(sk_k00);
      sk_k10 = __builtin_amdgcn_readfirstlane(sk_k10);
      sk_k11 = __builtin_amdgcn_readfirstlane(sk_k11);
    }
  }
  const int bx = __builtin_amdgcn_readfirstlane(kp.xcd ? (int)(blockIdx.x & 7u) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  auto unit_at = [&](int seq, int& item, int& k0, int& k1) -> bool {
    if (!SKC || kp.sk_tail == 0) {
      item = bx + seq * G;
      k0 = 0;
      k1 = -1;
      return item < kp.items;
    }
    const int u = seq - kp.sk_rf;
    if (u < 0) {
      item = (int)blockIdx.x + seq * G;
      k0 = 0;
      k1 = -1;
      return true;
    }
    if (u >= sk_n) return false;
    item = sk_item0 + u * kp.ngroups;
    k0 = u == 0 ? sk_k00 : 0;
    k1 = u == 0 ? sk_k10 : sk_k11;
    return true;
  };
  const i32x4 srdA = make_srd(p.in, kp.bytes_in);
  const i32x4 srdB = make_srd(p.wt, kp.bytes_wt);
  unsigned a_off[PA];    // byte offset of (row's pixel at tap offset (0,0)) + this lane's swizzled 16-byte chunk
  int a_h[PA], a_w[PA];  // i*IS, j*IS of the row (a_h very negative => row beyond the problem)
  unsigned b_off[PB];  // byte offset of this lane's weight row (n-tile 0, tap 0) + swizzled chunk
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = PB * 8 * wave + 8 * i + prow;
    b_off[i] = (unsigned)r * p.wtaps * p.Ck * ES + (unsigned)((pch ^ ((r >> 1) & 7)) * 16);
  }
  auto L_setup = [&]() {  // decode the rows this lane stages for unit L_seq; skips items without taps
    int uk0, uk1;
    while ((L_valid = unit_at(L_seq, L_item, uk0, uk1))) {
      const int rowtile = L_item / kp.ngroups;
      L_grp = L_item - rowtile * kp.ngroups;
      L_cls = rowtile / kp.mtiles;
      const int mt = rowtile - L_cls * kp.mtiles;
      const int nk_all = p.cls[L_cls].ntaps * kc_per_tap;
      if (nk_all == 0) {
        ++L_seq;
        continue;
      }
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = (BM / NW) * wave + 8 * i + prow;
        const int m = mt * BM + r;
        if (m < Msub) {
          const int j = m % p.Wsub;
          const int t = m / p.Wsub;
          const int ii = t % p.Hsub;
          const int n = t / p.Hsub;
          a_h[i] = ii * p.IS;
          a_w[i] = j * p.IS;
          const int chunk = pch ^ ((r >> 1) & 7);  // source 16-byte chunk of the 128-byte slab row
          a_off[i] = (unsigned)((n * p.Hin + a_h[i]) * p.Win + a_w[i]) * (unsigned)(p.pix_stride * ES) +
                     (unsigned)(chunk * 16);
          // stem row pairs: elements k >= 32 of a tap live in the next image row (bf16: chunks 4..7 of the slab;
          // fp32: the tap's second slab, see L_begin)
          if (ES == 2 && chunk >= 4) a_off[i] += (unsigned)(p.pair_delta * ES);
        } else {
          a_h[i] = -(1 << 20);
          a_w[i] = 0;
          a_off[i] = 0;
        }
      }
      L_nt = 0;
      L_kt = uk0;
      L_kend = uk1 < 0 ? nk_all : uk1;
      return;
    }
  };
  // per-slab uniform state of the loader, set by L_begin and consumed by L_piece
  int S_dh = 0, S_dw = 0, S_dA = 0;
  unsigned S_dB = 0, S_As = 0;
  auto L_begin = [&](int stage) {
    // the cursor is wave-uniform by construction; say so, or hipcc fetches the tap with a VECTOR load from the kernarg
    // segment and waits vmcnt(0) for it in the middle of the LDS-DMA burst
    const int kt_u = __builtin_amdgcn_readfirstlane(L_kt);
    const int cls_u = __builtin_amdgcn_readfirstlane(L_cls);
    const int t = kt_u / kc_per_tap;
    const int c0 = (kt_u - t * kc_per_tap) * BK;
    const Tap tp = p.cls[cls_u].taps[t];
    S_dh = tp.dh;
    S_dw = tp.dw;
    S_dA = ((tp.dh * p.Win + tp.dw) * p.pix_stride + c0 + (ES == 4 && c0 >= 32 ? p.pair_delta : 0)) * ES;
    const int n0 = __builtin_amdgcn_readfirstlane((L_grp * kp.ntpg + L_nt) * BN);
    S_dB = ((unsigned)n0 * p.wtaps + tp.wtap) * (unsigned)(p.Ck * ES) + (unsigned)(c0 * ES);
    S_As = __builtin_amdgcn_readfirstlane(lds_addr(smem) + stage * STAGE);
  };
  auto L_piece = [&](int j) {  // j < PA: A piece j;  j >= PA: B piece j-PA
    if (j < PA) {
      const int ih = a_h[j] + S_dh;
      const int iw = a_w[j] + S_dw;
      const bool ok = ((unsigned)ih < (unsigned)p.Hin) && ((unsigned)iw < (unsigned)p.Win);
      blds16(srdA, ok ? a_off[j] + (unsigned)S_dA : 0x80000000u, S_As + ((BM / NW) * wave + 8 * j) * BKB);
    } else {
      const int i = j - PA;
      blds16(srdB, b_off[i] + S_dB, S_As + A_BYTES + (PB * 8 * wave + 8 * i) * BKB);
    }
  };
  auto L_advance = [&]() {
    if (++L_kt < L_kend) return;
    if (++L_nt < kp.ntpg) {  // (never with stream-K units: ntpg == 1 there)
      L_kt = 0;
      return;
    }
    ++L_seq;
    L_setup();
  };

  // ---- fragment addresses (constant over the whole kernel) -------------------------------------------------------
  const int hh = lane >> 5;
  int a_row[MI], a_sw[MI], b_row[NI], b_sw[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = wm * (BM / WMW) + mi * 32 + (lane & 31);
    a_row[mi] = r * BKB;
    a_sw[mi] = (r >> 1) & 7;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * WN + ni * 32 + (lane & 31);
    b_row[ni] = A_BYTES + r * BKB;
    b_sw[ni] = (r >> 1) & 7;
  }

  // ---- prologue ---------------------------------------------------------------------------------------------------
  if constexpr (STATS != 0) {  // zeroed before the first barrier; first touched after the first epilogue barrier
    for (int i = tid; i < WMW * kp.ntpg * BN * 2; i += NT) stat_acc[i] = 0.f;
  }
  L_setup();
  int stage = 0;     // the ring stage the MFMAs read
  int L_stage = 0;   // the ring stage the loader fills next (NSTG - 1 slabs ahead)
  int inflight = 0;  // slabs issued and not yet consumed
#pragma unroll
  for (int pre = 0; pre < NSTG - 1; ++pre)
    if (L_valid) {
      L_begin(L_stage);
#pragma unroll
      for (int j = 0; j < NPC; ++j) L_piece(j);
      L_advance();
      L_stage = L_stage + 1 == NSTG ? 0 : L_stage + 1;
      ++inflight;
    }
  int pending_st = 0;
#ifdef MI355_STAMP
  int stamp_k = 0;
#endif

  T* out = reinterpret_cast<T*>(p.out);
  const T* addend = reinterpret_cast<const T*>(p.addend);

  int item, uk0, uk1;
  for (int seq = 0; unit_at(seq, item, uk0, uk1); ++seq) {
    const int rowtile = item / kp.ngroups;
    const int grp = item - rowtile * kp.ngroups;
    const int ci = rowtile / kp.mtiles;
    const int mt = rowtile - ci * kp.mtiles;
    const TapClass& cls = p.cls[ci];
    const int nk = cls.ntaps * kc_per_tap;
    const int m0 = mt * BM;
    for (int nti = 0; nti < kp.ntpg; ++nti) {
      const int n0 = (grp * kp.ntpg + nti) * BN;
      f32x16 acc[MI][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

      const int kbeg = uk0, kend = uk1 < 0 ? nk : uk1;
      for (int kt = kbeg; kt < kend; ++kt) {
        // the slab for this step was issued one step ago; epilogue stores issued since then may stay in flight
        STAMP(0);
        if constexpr (NSTG == 2) {
          if (pending_st == NST) {
            wait_vm<NST>();
          } else {
            wait_vm<0>();
          }
        } else {
          // the slab needed now is the OLDEST in flight; one younger slab (NPC pieces of this wave) and the epilogue
          // stores issued since may stay outstanding
          const bool younger = inflight >= 2;
          if (younger && pending_st == NST) wait_vm<NPC + NST>();
          else if (younger) wait_vm<NPC>();
          else if (pending_st == NST) wait_vm<NST>();
          else wait_vm<0>();
        }
        pending_st = 0;
        STAMP(1);
        MI355_LDS_BARRIER();  // slab landed for every wave; everyone is done reading the other stage
        STAMP(2);
        const bool lv = L_valid;
        if (lv) L_begin(L_stage);
        const char* base = smem + stage * STAGE;
        // Four 32-byte k groups per slab; lane half hh takes one 16-byte chunk of each.  The fragments of group g+1 are
        // read from LDS BEFORE the MFMAs of group g are issued (two register sets, static indices), and two of the next
        // slab's LDS-DMA pieces go out in between, so LDS latency and offset arithmetic hide under the matrix pipe.
        typedef typename std::conditional<ES == 4, f32x4, bf16x8>::type frag_t;
        frag_t av[2][MI], bv[2][NI];
        auto read_frags = [&](int g, int set) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            av[set][mi] = *reinterpret_cast<const frag_t*>(base + a_row[mi] + (((2 * g + hh) ^ a_sw[mi]) << 4));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            bv[set][ni] = *reinterpret_cast<const frag_t*>(base + b_row[ni] + (((2 * g + hh) ^ b_sw[ni]) << 4));
        };
        read_frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g + 1 < 4) read_frags(g + 1, (g + 1) & 1);
          if (lv) {
#pragma unroll
            for (int j = g * NPC / 4; j < (g + 1) * NPC / 4; ++j) L_piece(j);
          }
          if constexpr (ES == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[g & 1][ni][e], av[g & 1][mi][e], acc[mi][ni], 0, 0, 0);
          } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[g & 1][ni], av[g & 1][mi], acc[mi][ni], 0, 0, 0);
          }
        }
        STAMP(3);
        if (lv) {
          L_advance();
          L_stage = L_stage + 1 == NSTG ? 0 : L_stage + 1;
          ++inflight;
        }
        --inflight;
        STAMP(4);
#ifdef MI355_STAMP
        ++stamp_k;
#endif
        stage = stage + 1 == NSTG ? 0 : stage + 1;
      }

      // ---- stream-K: a partial tile is either handed to its owner or completed with the others' partials ----------
      if (SKC && kp.sk_tail != 0 && (kbeg > 0 || kend < nk)) {
        constexpr int SLOT = BM * BN;  // floats per workgroup slot, [MI*NI*4 register groups][256 threads][4]
        if (kbeg > 0) {
          // not the owner: write-through (sc1) stores, every wave drains them, one lane publishes the epoch
          const i32x4 srdP = make_srd(kp.sk_partial, (unsigned)((size_t)G * SLOT * 4));
          const unsigned base = ((unsigned)blockIdx.x * SLOT + tid * 4) * 4u;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                const unsigned off = base + (unsigned)(((mi * NI + ni) * 4 + g) * 256 * 16);
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1" ::"v"(v), "v"(off), "s"(srdP) : "memory");
              }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (tid == 0 && !kp.sk_mute)
            __hip_atomic_store(kp.sk_flags + blockIdx.x, kp.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pending_st = 0;
          continue;  // no epilogue for this unit
        }
        // owner: the rest of the tile's k range was cut into the ranges of the following workgroups of this group
        const unsigned TK = (unsigned)(kp.sk_tail * kp.sk_nk);
        const int tt = rowtile - (kp.sk_rf << kp.sk_gshift);   // tail tile index
        const unsigned tile_end = (unsigned)((tt + 1) * kp.sk_nk);
        int bgj = sk_bg + 1;
        int covered = kend;
        while (covered < nk && bgj < (1 << kp.sk_tshift)) {
          const unsigned lo_j = ((unsigned)bgj * TK) >> kp.sk_tshift, hi_j = ((unsigned)(bgj + 1) * TK) >> kp.sk_tshift;
          const int cnt = (int)((hi_j < tile_end ? hi_j : tile_end) - lo_j);
          if (cnt > 0) {
            const int j = bgj * kp.ngroups + grp;
            if (tid == 0) {
              unsigned spins = 0;
              while (__hip_atomic_load(kp.sk_flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kp.sk_epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kp.sk_spin_limit) {  // never on a healthy run; do not hang the device.  The tile is then
                  // completed WITHOUT that partial, i.e. wrong: the error word makes the executor fail the step
                  __hip_atomic_store(kp.sk_flags + IGEMM_SK_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  break;
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float* slot = kp.sk_partial + (size_t)j * SLOT + tid * 4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  const f32x4 v = *reinterpret_cast<const f32x4*>(slot + ((mi * NI + ni) * 4 + g) * 256 * 4);
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[mi][ni][4 * g + e] += v[e];
                }
            covered += cnt;
            pending_st = 0;  // the wait above drained every store
          }
          ++bgj;
        }
      }

      // ---- epilogue: the slab buffer just consumed is free once every wave has passed the barrier -------------------
      const int done_stage = stage == 0 ? NSTG - 1 : stage - 1;
      char* stg = smem + done_stage * STAGE + (wave % EPI_WAVES) * STG_WAVE;
      MI355_LDS_BARRIER();
      if (MI355_PROBE(1)) continue;  // timing probe: results are not written
      if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1;
        if (m < Msub) {
          const int j = m % p.Wsub;
          const int t = m / p.Wsub;
          const int ii = t % p.Hsub;
          const int n = t / p.Hsub;
          pix = (n * p.Hout + ii * p.OS + cls.ph) * p.Wout + j * p.OS + cls.pw;
        }
        row_pix[tid] = pix;
      }
      if constexpr (EPI_TURNS > 1) MI355_LDS_BARRIER();  // row_pix visible (the turns below have no barrier inside)
#pragma unroll
      for (int turn = 0; turn < EPI_TURNS; ++turn) {
        if (EPI_TURNS == 1 || wave / EPI_WAVES == turn) {
        const int prw = lane & 31;
        const int rr = lane / CPR, ch = lane % CPR;
        // STATS 1: this lane's share of sum / sum of squares of the tile's outputs AS STORED
        // STATS 2: of sum dz / sum dz*xhat, dz = stored output under the ReLU mask bn_bits, xhat from bn_y
        float s1[VEC], s2[VEC];
        float bmu[VEC], bis[VEC];
        if constexpr (STATS == 2) {
          const int c0 = n0 + wn * WN + ch * VEC;
  #pragma unroll
          for (int q = 0; q < VEC / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.bn_mean + c0 + 4 * q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bn_invstd + c0 + 4 * q);
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              bmu[4 * q + e] = a[e];
              bis[4 * q + e] = b[e];
            }
          }
        }
  #pragma unroll
        for (int e = 0; e < VEC; ++e) {
          s1[e] = 0.f;
          s2[e] = 0.f;
        }
  #pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          // a lane holds pixel prw and, per register group g, channels ni*32 + 8g + 4hh + {0..3}
  #pragma unroll
          for (int ni = 0; ni < NI; ++ni)
  #pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
              const int sc = (ni * 8 + 2 * g + hh) ^ (prw & (SCH - 1));
              *reinterpret_cast<f32x4*>(stg + prw * (WN * 4) + sc * 16) = v;
            }
          if (mi == 0 && EPI_TURNS == 1) {
            MI355_LDS_BARRIER();  // row_pix visible to all waves (the staging region itself is wave-private)
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          constexpr int NP = 32 / RPI;
          int pixs[NP];
          uint4 araw[NP];
          unsigned abits[NP];
          uint4 yraw[NP];
          unsigned ybits[NP];
  #pragma unroll
          for (int ps = 0; ps < NP; ++ps) {  // all addend loads of this pass first: one round trip, not NP
            pixs[ps] = row_pix[wm * (BM / WMW) + mi * 32 + ps * RPI + rr];
            if (addend) {
              const size_t o = (size_t)(pixs[ps] < 0 ? 0 : pixs[ps]) * p.Ncols + n0 + wn * WN + ch * VEC;
              araw[ps] = *reinterpret_cast<const uint4*>((pixs[ps] < 0 || MI355_PROBE(4)) ? reinterpret_cast<const T*>(g_zero_page) : addend + o);
              abits[ps] = p.addend_bits ? (unsigned)p.addend_bits[pixs[ps] < 0 ? 0 : o / VEC] : 0xffu;
            }
            if constexpr (STATS == 2) {
              const size_t o = (size_t)(pixs[ps] < 0 ? 0 : pixs[ps]) * p.Ncols + n0 + wn * WN + ch * VEC;
              yraw[ps] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.bn_y) + o);
              ybits[ps] = (unsigned)p.bn_bits[o / VEC];
            }
          }
  #pragma unroll
          for (int ps = 0; ps < NP; ++ps) {
            const int row = ps * RPI + rr;
            const int pix = pixs[ps];
            float v[VEC];
  #pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
              const int sc = (ch * (VEC / 4) + q) ^ (row & (SCH - 1));
              const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * (WN * 4) + sc * 16);
              v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
            }
            const size_t o = (size_t)(pix < 0 ? 0 : pix) * p.Ncols + n0 + wn * WN + ch * VEC;
            if (addend) {
              float a[VEC];
              Vec16<T>::unpack(araw[ps], a);
  #pragma unroll
              for (int e = 0; e < VEC; ++e) v[e] += (abits[ps] >> e) & 1u ? a[e] : 0.f;
            }
            // rows past the end of the problem go to a trash page so that every thread issues exactly NST stores
            T* dst = (pix < 0 || MI355_PROBE(2)) ? reinterpret_cast<T*>(g_trash + tid * 16) : out + o;
            Vec16<T>::store(dst, v);
            if constexpr (STATS == 1) {
              if (pix >= 0) {
  #pragma unroll
                for (int e = 0; e < VEC; ++e) {
                  const float xr = (float)(T)v[e];
                  s1[e] += xr;
                  s2[e] += xr * xr;
                }
              }
            }
            if constexpr (STATS == 2) {
              if (pix >= 0) {
                float yv[VEC];
                Vec16<T>::unpack(yraw[ps], yv);
  #pragma unroll
                for (int e = 0; e < VEC; ++e) {
                  const float dz = (ybits[ps] >> e) & 1u ? (float)(T)v[e] : 0.f;
                  s1[e] += dz;
                  s2[e] += dz * ((yv[e] - bmu[e]) * bis[e]);
                }
              }
            }
          }
          asm volatile("" ::: "memory");
        }
        if constexpr (STATS != 0) {
          // lanes with the same channel chunk sit CPR lanes apart (rows rr): park the per-lane sums in the wave-private
          // staging region, let lane c add up channel c's RPI rows and add the result to THIS wave's accumulator slot
          // (one fixed lane per slot, LDS is in-order per wave => deterministic, no atomics, nothing leaves the CU).
          float* scr = reinterpret_cast<float*>(stg);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  #pragma unroll
          for (int q = 0; q < VEC / 4; ++q) {
            const f32x4 a = {s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]};
            const f32x4 b = {s2[4 * q], s2[4 * q + 1], s2[4 * q + 2], s2[4 * q + 3]};
            *reinterpret_cast<f32x4*>(scr + rr * WN + ch * VEC + 4 * q) = a;
            *reinterpret_cast<f32x4*>(scr + (RPI + rr) * WN + ch * VEC + 4 * q) = b;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  #pragma unroll
          for (int cc = lane; cc < WN; cc += 64) {
            float a = 0.f, b = 0.f;
  #pragma unroll
            for (int r = 0; r < RPI; ++r) {
              a += scr[r * WN + cc];
              b += scr[(RPI + r) * WN + cc];
            }
            float* slot = stat_acc + ((wm * kp.ntpg + nti) * BN + wn * WN + cc) * 2;
            slot[0] += a;
            slot[1] += b;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        }
        if constexpr (EPI_TURNS > 1) MI355_LDS_BARRIER();  // the next group of waves takes the staging blocks
      }
      pending_st += NST;
    }
  }
  if constexpr (STATS != 0) {
    // flush: partial[workgroup / ngroups][0|1][channel], the two wave rows added.  Every item of a workgroup has the
    // same n-tile group because gridDim.x is a multiple of ngroups, so the workgroup owns channels
    // [grp*chan, (grp+1)*chan) and the ngroups workgroups bx/ngroups == r fill row r completely (bx: blockIdx.x or its XCD remap).
    MI355_LDS_BARRIER();
    const int chan = kp.ntpg * BN;
    const int grp = bx % kp.ngroups;
    float* row = p.stat_partial + (size_t)(bx / kp.ngroups) * 2 * p.Ncols + grp * chan;
    for (int c = tid; c < chan; c += NT) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WMW; ++w) {  // the wave rows, in order
        a += stat_acc[(w * chan + c) * 2];
        b += stat_acc[(w * chan + c) * 2 + 1];
      }
      row[c] = a;
      row[p.Ncols + c] = b;
    }
  }
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel symbol (once)
void lds_opt_in(const void* fn, size_t lds) {
  if (lds <= 64 * 1024) return;
  static std::mutex mu;
  static std::set<const void*> done;
  std::lock_guard<std::mutex> g(mu);
  if (done.count(fn)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.insert(fn);
}

template <typename T, int BM, int BN, int WMW = 2, int NSTG = 2>
int launch_t(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows) {
  constexpr int NT = 128 * WMW;
  const int MAX_WG = device_cus() * (BM == 256 ? 1 : (BN == 64 && NSTG == 2 ? 3 : 2));  // persistent workgroups: 1, 3 or 2 per CU
  IgemmKArgs k;
  k.a = a;
  const int Msub = a.N * a.Hsub * a.Wsub;
  k.mtiles = cdiv(Msub, BM);
  k.ny = a.Ncols / BN;
  k.nclass = nclass;
  const int R = nclass * k.mtiles;
  // n-tile groups: a workgroup item = one row tile x (ny / ng) n-tiles.
  //   bf16: the smallest ng that fills the slots — more n-tiles per item keep the A rows hot, which is what the
  //         memory-leaning bf16 layers want (the round-count rule below measured +0.2 ms/step there);
  //   fp32: compute-bound, so the split with the fewest tile-rounds ceil(R * ng / slots) * (ny / ng) (-0.4 ms/step).
  int ng = k.ny;
  if (sizeof(T) == 4) {
    long best_rounds = -1;
    for (int d = 1; d <= k.ny; ++d) {
      if (k.ny % d != 0) continue;
      const long rounds = (long)cdiv(R * d, MAX_WG) * (k.ny / d);
      if (best_rounds < 0 || rounds < best_rounds) {
        best_rounds = rounds;
        ng = d;
      }
    }
  } else {
    for (int d = 1; d <= k.ny; ++d)
      if (k.ny % d == 0 && R * d >= MAX_WG) {
        ng = d;
        break;
      }
  }
  // stream-K over the partial last round: one n-tile per item (ng = ny), and only where it pays — a tail that leaves
  // >= 15 % of the slots idle and a reduction long enough to cut (see IgemmKArgs)
  k.sk_rf = k.sk_tail = k.sk_nk = k.sk_gshift = k.sk_tshift = 0;
  k.sk_epoch = 0;
  k.sk_partial = nullptr;
  k.sk_flags = nullptr;
  k.sk_mute = knobs().sk_mute ? 1 : 0;   // (MI355_SK_DEBUG=mute: the test hook of the stream-K failure path)
#ifdef MI355_PROBES
  k.dbg = probe_env("MI355_IGEMM_DBG");  // read per launch: timing probes only, results are wrong with it set
#else
  k.dbg = 0;
#endif
  k.sk_spin_limit = k.sk_mute ? (1u << 8) : (1u << 24);
  bool sk = false;
  // (fp32 only: a hand-off costs the owner ~10 us, nothing beside a 150-300 us fp32 tile, but most of what the cut
  // saves on a 30-60 us bf16 tile — measured)
  if (BM == 128 && BN == 128 && sizeof(T) == 4 && a.sk_ws && nclass == 1 && MAX_WG % k.ny == 0) {
    const int Gg = MAX_WG / k.ny;
    const int rf = R / Gg, rt = R - rf * Gg;
    const int nk = a.cls[0].ntaps * (a.Ck / (BKB / (int)sizeof(T)));
    if (rt > 0 && rt * 100 <= Gg * 85 && nk >= 8 && (Gg & (Gg - 1)) == 0 && (long)rt * nk * MAX_WG < (1l << 31)) {
      sk = true;
      ng = k.ny;
      k.sk_rf = rf;
      k.sk_tail = rt;
      k.sk_nk = nk;
      k.sk_gshift = 0;
      while ((1 << k.sk_gshift) < Gg) ++k.sk_gshift;
      k.sk_tshift = 0;  // largest power of two <= min(Gg, 4 * rt)
      while ((2 << k.sk_tshift) <= Gg && (2 << k.sk_tshift) <= 4 * rt) ++k.sk_tshift;
      static std::atomic<unsigned> epoch{0};
      k.sk_epoch = ++epoch;
      if (k.sk_epoch == 0) k.sk_epoch = ++epoch;  // 0 is the initial state of the flags
      k.sk_flags = reinterpret_cast<unsigned*>(a.sk_ws);
      k.sk_partial = reinterpret_cast<float*>(reinterpret_cast<char*>(a.sk_ws) + IGEMM_SK_FLAG_BYTES);
    }
  }
  // Epilogues that read ReLU bit masks (1 byte per 16-byte vector, [pixel][Ncols / 8]: the shortcut addend's mask and the
  // BN-backward mask of conv1's dgrad) touch 8 bytes of a 128-byte line per wave tile; with several n-tiles per item the line has
  // left the L2 again before the next n-tile asks for it, and the other n-tile groups sit behind other XCDs' L2s: PMC shows the
  // mask bytes fetched up to 16x (profiles/r03_pmc_mask_amplification.txt: 97 MB for a 6.4 MB mask).  One n-tile per item + the XCD
  // remap puts all n-tiles of a row tile behind ONE L2 at the same time (and its A rows are fetched once instead of ng times):
  // conv1's dgrad of layers 2-4 612 -> 366 MB and 112 -> 99 us (layer 3), 417 -> 226 MB and 84 -> 67 us (layer 4), 156 -> 146 us
  // (layer 2); the 256-column launches of layer 1 (2 n-tiles) lose 8 us and keep the plain order.
  k.xcd = 0;
  if (sizeof(T) == 2 && !sk && BM == 128 && MAX_WG % 8 == 0 && (MAX_WG / 8) % k.ny == 0 && R * k.ny >= MAX_WG && k.ny >= 2 &&
      (a.addend_bits != nullptr || a.bn_bits != nullptr) && a.Ncols >= 512) {
    ng = k.ny;
    k.xcd = 1;
  }
  k.ngroups = ng;
  k.ntpg = k.ny / ng;
  k.items = R * ng;
  const size_t bytes_in = (size_t)a.N * a.Hin * a.Win * a.pix_stride * sizeof(T);
  const size_t bytes_wt = (size_t)a.Ncols * a.wtaps * a.Ck * sizeof(T);
  MI355_ARG(bytes_in < 0x80000000ull && bytes_wt < 0x80000000ull, "igemm: tensor exceeds the 2 GiB buffer-offset range");
  k.bytes_in = (unsigned)bytes_in;
  k.bytes_wt = (unsigned)bytes_wt;
  const int grid = sk ? MAX_WG : (k.items < MAX_WG ? k.items : MAX_WG);
  size_t lds = (size_t)NSTG * (BM + BN) * BKB + BM * sizeof(int);
  // BN statistics in the epilogue need [2][channels per workgroup][2] floats of LDS; beyond 512 channels per workgroup
  // the kernel would drop to one workgroup per CU, so the caller falls back to the standalone statistics kernel
  const int chan = k.ntpg * BN;
  const bool stats = a.stat_partial != nullptr && chan <= 512 && grid % ng == 0 &&
                     lds + (size_t)WMW * chan * 2 * sizeof(float) <= 160 * 1024;
  if (stat_rows) *stat_rows = stats ? grid / ng : 0;
  if (stats) {
    lds += (size_t)WMW * chan * 2 * sizeof(float);
    lds_opt_in((const void*)igemm_kernel<T, BM, BN, WMW, NSTG, 1>, lds);
    lds_opt_in((const void*)igemm_kernel<T, BM, BN, WMW, NSTG, 2>, lds);
    if (a.bn_y)
      hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WMW, NSTG, 2>), dim3(grid), dim3(NT), lds, stream, k);
    else
      hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WMW, NSTG, 1>), dim3(grid), dim3(NT), lds, stream, k);
  } else {
    lds_opt_in((const void*)igemm_kernel<T, BM, BN, WMW, NSTG, 0>, lds);
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WMW, NSTG, 0>), dim3(grid), dim3(NT), lds, stream, k);
  }
  MI355_LAUNCH_CHECK();
  note_kernel("igemm<%s,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, NSTG);
  return 0;
}

}  // namespace

size_t igemm_sk_ws_bytes() { return IGEMM_SK_FLAG_BYTES + (size_t)512 * 128 * 128 * sizeof(float); }

// stem_direct.hip: the stem as a direct convolution
bool stem_direct_legal(const IgemmArgs& a, int nclass);
int launch_stem_direct(const IgemmArgs& a, hipStream_t stream, int* stat_rows);

// dconv.cpp: the generated one-wave-per-SIMD direct 3x3 / stride-1 kernels (asm/dconv_gen.py) for the layer-3 / layer-4 shapes
bool dconv_legal(const IgemmArgs& a, int nclass);
int launch_dconv(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows);
bool pw_legal(const IgemmArgs& a, int nclass);  // the persistent pointwise kernels (asm/pw_gen.py): output-heavy 1x1 forward
int launch_pw(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows);
bool pk_legal(const IgemmArgs& a, int nclass);  // the long-reduction pointwise kernels (asm/pk_gen.py): K = 1024 / 2048 -> 256-column tiles
int launch_pk(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows);
bool po_legal(const IgemmArgs& a, int nclass);  // the output-heavy pointwise kernels with resident weights (asm/po_gen.py): K <= 512 -> 4K columns, shortcut addend, BN-backward sums
int launch_po(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows);

// conv_igemm8.hip
bool igemm8_legal(const IgemmArgs& a, int nclass, int bn);
int launch_igemm8(const IgemmArgs& a, int nclass, int bm, int bn, int korder, int fat, hipStream_t stream, int* stat_rows);

// Which bf16 launches go to the 8-wave ping-pong kernel, and with which tile.  MI355_IGEMM8 in the environment:
//   "0" never;  "<BM>x<BN>[k][f]" (e.g. 256x256, 224x128kf) forces that tile wherever it is legal (k: channel chunks
//   outer, taps inner; f: the fat-phase form);  unset: the measured rule below.
static bool choose_igemm8(const IgemmArgs& a, int nclass, int* bm, int* bn, int* korder, int* fat) {
  const char* env = knobs().has_igemm8 ? knobs().igemm8 : nullptr;
  if (env && env[0] == '0') return false;
  if (env && env[0]) {
    int m = 0, n = 0;
    char k1 = 0, k2 = 0;
    if (sscanf(env, "%dx%d%c%c", &m, &n, &k1, &k2) >= 2 && (m == 256 || m == 224) && (n == 256 || n == 128) && igemm8_legal(a, nclass, n)) {
      *bm = m; *bn = n; *korder = k1 == 'k' || k2 == 'k'; *fat = k1 == 'f' || k2 == 'f';
      return true;
    }
    return false;
  }
  // Measured per layer shape at batch 256 (tools/conv8_check.py, same-process A/B against the 4-wave tiles):
  //  - >= 256 output columns and a reduction of >= 256: the 224 x 256 tile wins on every layer-3/4 shape as long as its
  //    tile count still covers most of the 256 CUs (it is bound by fragment reads + LDS-DMA issue, not by MFMAs, and a
  //    224-row tile has 1/8 fewer A reads than a 256-row one; 224 divides the 49 * 2^k * N pixel counts);
  //  - the 512-column layer-4 3x3 (98 tiles of 256 x 256, 112 of 224 x 256): 256 x 128 fat phases, 196 tiles.
  int max_taps = 0;
  for (int ci = 0; ci < nclass; ++ci) max_taps = a.cls[ci].ntaps > max_taps ? a.cls[ci].ntaps : max_taps;
  const long K = (long)max_taps * a.Ck;
  const long M = (long)a.N * a.Hsub * a.Wsub;
  //  - NOT the output-heavy launches: one workgroup per CU runs its epilogue with the matrix pipe idle, so a short
  //    reduction under a long epilogue (conv1's dgrad: K = 256 / 512 into 1024 / 2048 columns, + shortcut addend + the
  //    BN-backward sums) loses 10-65 us per launch against two independent 4-wave workgroups per CU, and so does a
  //    multi-round launch of short tap classes (the stride-2 3x3 dgrad of layer 3) — measured in the executor,
  //    profiles/r02a_conv_per_layer_bf16_serial_{old,rule}.txt.
  const bool heavy_epilogue = (a.addend != nullptr && K < 1024);
  // k order of a multi-tap launch: channel chunks outer, taps inner — the 9 taps of a 64-channel chunk re-read the same
  // A rows back to back, so they are served from the XCD's L2 instead of being fetched again from beyond it (layer-4 3x3:
  // 373 -> see 77 MB per launch for 25.7 MB of activations, profiles/r02c_pmc_per_conv_launch_bf16_serial.txt; same speed)
  const int ko = (max_taps > 1 && a.Ck > 64) ? 1 : 0;
  if (a.Ncols % 256 == 0 && K >= 256 && !heavy_epilogue && igemm8_legal(a, nclass, 256)) {
    const long tiles = ((M + 223) / 224) * nclass * (a.Ncols / 256);
    const int cus = device_cus();  // thresholds measured on 256 CUs, kept as fractions of the chip (0.7 of a round; two rounds)
    if (tiles * 10 >= 7L * cus && !(nclass > 1 && tiles > 2L * cus && max_taps > 1)) {
      *bm = 224; *bn = 256; *korder = ko; *fat = 0;
      return true;
    }
  }
  if (a.Ncols % 128 == 0 && K >= 4096 && igemm8_legal(a, nclass, 128)) {
    const long tiles = ((M + 255) / 256) * nclass * (a.Ncols / 128);
    if (tiles * 10 >= 7L * device_cus() && tiles <= device_cus()) {
      *bm = 256; *bn = 128; *korder = ko; *fat = 1;
      return true;
    }
  }
  return false;
}

bool igemm_leaky_sums_legal(int dtype, const IgemmArgs& a, int nclass) {
  if (dtype != MI355_BF16 || !a.bn_y || !a.stat_partial || a.bn_slope != 0.01f || knobs().error[0]) return false;
  return dconv_legal(a, nclass) || po_legal(a, nclass) || pk_legal(a, nclass);
}

int launch_igemm(int dtype, const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows) {
  const int bk = BKB / (int)dtype_size(dtype);
  MI355_ARG(a.in && a.wt && a.out, "igemm: null pointer");
  MI355_ARG(a.Ck % bk == 0, "igemm: Ck=%d not a multiple of %d", a.Ck, bk);
  MI355_ARG(a.Ncols % 64 == 0, "igemm: Ncols=%d not a multiple of 64", a.Ncols);
  MI355_ARG(nclass >= 1 && nclass <= 4, "igemm: nclass=%d", nclass);
  MI355_ARG(((size_t)a.pix_stride * dtype_size(dtype)) % 8 == 0, "igemm: pixel stride not 8-byte aligned");
  MI355_ARG(a.N > 0 && a.Hsub > 0 && a.Wsub > 0, "igemm: empty problem");
  MI355_ARG(!knobs().error[0], "%s", knobs().error);
  // BN = 128 unless that leaves most of the 256 CUs without a tile (the FC layer: 256 rows): then 64-wide tiles double
  // the workgroups
  const long tiles128 = (long)cdiv(a.N * a.Hsub * a.Wsub, 128) * nclass * (a.Ncols / 128);
  const bool wide = (a.Ncols % 128 == 0) && tiles128 * 2 >= device_cus();
  MI355_ARG(dtype == MI355_BF16 || !a.addend_sub2, "igemm: a half-resolution addend needs the generated pointwise kernel (igemm_sub2_legal)");
  MI355_ARG(a.bn_slope == 0.f || igemm_leaky_sums_legal(dtype, a, nclass), "igemm: BN-backward sums under a leaky mask need a generated kernel with that epilogue and slope 0.01 (igemm_leaky_sums_legal)");
  MI355_ARG(a.bn_in == nullptr || igemm_bn_in_legal(dtype, a, nclass), "igemm: the input's BatchNorm in the operand path needs a generated kernel with that form (igemm_bn_in_legal)");
  if (dtype == MI355_F32)
    return wide ? launch_t<float, 128, 128>(a, nclass, stream, stat_rows) : launch_t<float, 128, 64>(a, nclass, stream, stat_rows);
  if (dtype == MI355_BF16) {
    if (dconv_legal(a, nclass)) return launch_dconv(a, nclass, stream, stat_rows);
    if (po_legal(a, nclass)) return launch_po(a, nclass, stream, stat_rows);
    MI355_ARG(!a.addend_sub2, "igemm: a half-resolution addend needs the generated pointwise kernel (igemm_sub2_legal)");
    if (pw_legal(a, nclass)) return launch_pw(a, nclass, stream, stat_rows);
    if (pk_legal(a, nclass)) return launch_pk(a, nclass, stream, stat_rows);
    {
      // the stem as a direct convolution out of raw input rows (stem_direct.hip; MI355_STEM_DIRECT=0: the row-pair implicit GEMM)
      if (knobs().stem_direct && stem_direct_legal(a, nclass)) return launch_stem_direct(a, stream, stat_rows);
    }
    {
      int bm8 = 0, bn8 = 0, ko8 = 0, fat8 = 0;
      if (choose_igemm8(a, nclass, &bm8, &bn8, &ko8, &fat8)) return launch_igemm8(a, nclass, bm8, bn8, ko8, fat8, stream, stat_rows);
    }
    // 256 x 256 tiles (bf16 only: fp32 MFMAs are slow enough that the LDS port is not the limit).  Measured per layer
    // shape at batch 256 (tools/one_conv.py): they win when the 256 single-workgroup CUs are still mostly filled
    // (>= 192 tiles) and the reduction is long enough to amortise the larger epilogue (K >= 256); they lose on the
    // HBM-bound layer-1/2 shapes and when layer 4's 98 row tiles leave most CUs idle.
    const int big_mode = knobs().has_igemm_big ? knobs().igemm_big : -1;  // MI355_IGEMM_BIG: 0 never / 1 wherever N % 256 == 0 (tests, A/B); unset: the rule
    const long items256 = (long)cdiv(a.N * a.Hsub * a.Wsub, 256) * nclass * (a.Ncols / 256);
    int max_taps = 0;
    for (int ci = 0; ci < nclass; ++ci) max_taps = a.cls[ci].ntaps > max_taps ? a.cls[ci].ntaps : max_taps;
    // (not with the BN-backward sums: that epilogue needs more registers than the 256 x 256 tile leaves)
    const int cus = device_cus();
    const bool big = a.Ncols % 256 == 0 && (big_mode < 0 ? (items256 * 4 >= 3L * cus && max_taps * a.Ck >= 256 && !a.bn_y) : big_mode == 1);
    if (big) return launch_t<bf16_t, 256, 256>(a, nclass, stream, stat_rows);
    // 256 x 128, 8 waves, 3-stage ring: per CU and k-step 8 % faster than two 128 x 128 workgroups (the slab wait drops
    // from ~700 to ~200 cycles), but a partial round costs it a full one where the 2-workgroup form speeds up when a CU
    // holds a single workgroup — so only where all its tiles fit into one round, and the reduction is long
    const long items3 = (long)cdiv(a.N * a.Hsub * a.Wsub, 256) * nclass * (a.Ncols / 128);
    const bool tall = a.Ncols % 128 == 0 &&
                      (big_mode < 0 ? ((items3 <= cus && items3 * 2 >= cus && max_taps * a.Ck >= 512) || (items3 <= 2L * cus && max_taps == 1 && a.Ck >= 1024)) : big_mode == 3);
    if (tall) return launch_t<bf16_t, 256, 128, 4, 3>(a, nclass, stream, stat_rows);
    return wide ? launch_t<bf16_t, 128, 128>(a, nclass, stream, stat_rows) : launch_t<bf16_t, 128, 64>(a, nclass, stream, stat_rows);
  }
  set_error("igemm: bad dtype %d", dtype);
  return MI355_E_ARG;
}

}  // namespace mi355

#ifdef MI355_STAMP
extern "C" int mi355_debug_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mi355::g_stamps), (size_t)n * sizeof(unsigned long long), 0,
                                  hipMemcpyDeviceToHost);
}
extern "C" int mi355_debug_stamps_clear(void) {
  static unsigned long long z[8 * 4096];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi355::g_stamps), z, sizeof(z), 0, hipMemcpyHostToDevice);
}
#endif
