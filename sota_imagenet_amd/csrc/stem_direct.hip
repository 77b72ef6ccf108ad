// stem_direct.hip — the 7x7 / stride-2 stem of ResNet-50 as a DIRECT bf16 convolution for gfx950 (MI355X): cuDNN's conv forward of
// the first layer under `model(data)` (/root/reference/sota_imagenet/callbacks.py:316).  Same contract as conv_igemm.hip (IgemmArgs;
// BN statistics in the epilogue); launch_igemm() routes the stem here.  The padded NHWC4 input rows of a tile are
// copied to LDS unchanged (contiguous LDS-DMA pieces, double buffered) and the MFMA pixel fragment of image row kh is read straight
// out of them; weights stay in LDS; two workgroups per CU.
// (Round 3's direct 3x3 kernel with DPP tap reuse, which shared this file and measured no faster than the implicit GEMM, lives in
// tools/experiments/conv3x3_dpp_round3.hip with its measurements; it is not part of the library any more.)
#include <algorithm>
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

constexpr int C3 = 64;             // channels on both sides
constexpr int NT3 = 4;              // 16-channel tiles per wave

__device__ __attribute__((aligned(256))) unsigned char g3_trash[256 * 16];

struct Conv3KArgs {
  IgemmArgs a;
  int TH, WP, F;            // output rows per tile, padded row pitch (pixels), 16-pixel fragments per tile (TH * WP / 16)
  int tiles_per_img, tiles;
  int npix;                 // pixels of the LDS image: (TH + 2) * WP + 2
  int npix_alloc;           // ... allocated: the 4 x 128 positions the waves always compute, + halo (reads behind the tile stay in LDS)
  unsigned bytes_in, bytes_wt;
  unsigned magWP;           // floor(2^32 / WP) + 1 (exact quotients for the < 2^16 positions of a tile)
  int wtap[9];              // weight tap index of (dh + 1) * 3 + (dw + 1)
  int dbg;                  // MI355_PROBES builds (MI355_CONV3_DBG): 1 skip the MFMA loop, 2 skip the epilogue, 4 no prefetch / commit after the first tile
};
#ifdef MI355_PROBES
#define C3_PROBE(bit) ((kp.dbg & (bit)) != 0)
#else
#define C3_PROBE(bit) false
#endif

void lds_opt_in3(const void* fn) {
  static std::mutex mu;
  static std::set<const void*> done;
  std::lock_guard<std::mutex> g(mu);
  if (done.count(fn)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.insert(fn);
}

template <int CTRL>
__device__ __forceinline__ float row_shr_add3(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true);
  return x + __int_as_float(y);
}
__device__ __forceinline__ float row_sum16_3(float x) {  // lane 15 of every 16-lane row ends with the row's sum (fixed order)
  x = row_shr_add3<0x111>(x);
  x = row_shr_add3<0x112>(x);
  x = row_shr_add3<0x114>(x);
  x = row_shr_add3<0x118>(x);
  return x;
}

typedef int i32x4v __attribute__((ext_vector_type(4)));

// fragment moved one lane DOWN the 16-lane rows (lane r takes lane r + 1; lane 15 takes lane 0 of `next`): the operand of tap dw = +1
__device__ __forceinline__ bf16x8 shift_next(bf16x8 cur, bf16x8 next) {
  const i32x4v c = __builtin_bit_cast(i32x4v, cur), n = __builtin_bit_cast(i32x4v, next);
  i32x4v r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = __builtin_amdgcn_update_dpp(0, n[i], 0x11F, 0xf, 0xf, true);      // row_shr:15 — lane 15 <- next lane 0 (others: 0, overwritten)
    r[i] = __builtin_amdgcn_update_dpp(t, c[i], 0x101, 0xf, 0xf, false);            // row_shl:1  — lanes 0..14 <- cur lane + 1
  }
  return __builtin_bit_cast(bf16x8, r);
}
// ... one lane UP (lane r takes lane r - 1; lane 0 takes lane 15 of `prev`): the operand of tap dw = -1
__device__ __forceinline__ bf16x8 shift_prev(bf16x8 cur, bf16x8 prev) {
  const i32x4v c = __builtin_bit_cast(i32x4v, cur), p = __builtin_bit_cast(i32x4v, prev);
  i32x4v r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = __builtin_amdgcn_update_dpp(0, p[i], 0x10F, 0xf, 0xf, true);      // row_shl:15 — lane 0 <- prev lane 15 (others: 0, overwritten)
    r[i] = __builtin_amdgcn_update_dpp(t, c[i], 0x111, 0xf, 0xf, false);            // row_shr:1  — lanes 1..15 <- cur lane - 1
  }
  return __builtin_bit_cast(bf16x8, r);
}

// epilogue of one tile, straight from the accumulators (shared by both kernel forms): output pixel of fragment mt = tile_pix0 + orel[mt]
// (orel < 0: padding column / behind the tile -> dropped)
template <int STATS, int MT>
__device__ __forceinline__ void conv3_epilogue(const Conv3KArgs& kp, f32x4 (&acc)[MT][NT3], const int (&orel)[MT], int tile, float* stat_acc, int wave,
                                               int px, int q4, int tid) {
  const IgemmArgs& p = kp.a;
  const int H = p.Hout, W = p.Wout;
  bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* addend = reinterpret_cast<const bf16_t*>(p.addend);
  constexpr int MT3 = MT;  // (the body below is written in terms of MT3)
  const int n_img = tile / kp.tiles_per_img;
  const int r0 = (tile - n_img * kp.tiles_per_img) * kp.TH;
  const int tile_pix0 = (n_img * H + r0) * W;
  const int cb = q4 * 8;  // this lane's 8 channels inside a pair of 16-row tiles (weight rows are permuted accordingly)
#pragma unroll
  for (int ntp = 0; ntp < NT3 / 2; ++ntp) {
    const int c0 = ntp * 32 + cb;
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
    constexpr int MB = 4;
#pragma unroll
    for (int mb = 0; mb < MT3; mb += MB) {
      int pixs[MB];
      uint4 araw[MB], yraw[MB];
      unsigned abits[MB], ybits[MB];
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        const int mt = mb + u;
        const int pix = orel[mt] < 0 ? -1 : tile_pix0 + orel[mt];
        pixs[u] = pix;
        const size_t o = (size_t)(pix < 0 ? 0 : pix) * C3 + c0;
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
        const f32x4 a = acc[mt][2 * ntp], b = acc[mt][2 * ntp + 1];
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = a[e];
          v[4 + e] = b[e];
        }
        const int pix = pixs[u];
        if (addend) {
          float ad[8];
          Vec16<bf16_t>::unpack(araw[u], ad);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (pix >= 0 && ((abits[u] >> e) & 1u)) ? ad[e] : 0.f;
        }
        const size_t o = (size_t)(pix < 0 ? 0 : pix) * C3 + c0;
        bf16_t* dst = pix < 0 ? reinterpret_cast<bf16_t*>(g3_trash + tid * 16) : out + o;
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
      float* slot = stat_acc + (wave * C3 + c0) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = row_sum16_3(s1[e]);
        const float b = row_sum16_3(s2[e]);
        if (px == 15) {
          slot[2 * e] += a;
          slot[2 * e + 1] += b;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The 7x7 / stride-2 stem (3 input channels, stored as zero-padded NHWC4 rows by stem_ingest) as a direct convolution.
//
// The implicit-GEMM form (conv_igemm.hip, 4 row-pair taps of 64) stages 512 bytes of gathered input per output pixel — neighbouring
// output columns overlap in 6 of their 8 input pixels — in 64-byte LDS-DMA pieces: 1.6 GB of staging per launch at batch 256 /
// 224 px, and the launch takes 256 us for 0.5 GB of HBM traffic.  Here the padded input rows a tile needs are copied to LDS AS THEY
// ARE (contiguous in memory: 2 TH + 5 rows for TH output rows, 25 KB) and the MFMA pixel fragment of image row kh is read straight
// out of them: output column ow needs input pixels 2 ow ... 2 ow + 7 of that row = 64 contiguous bytes, of which lane (ow, q) takes
// the 16 bytes (2 pixels x 4 channels) of k-group q — 16-byte aligned, consecutive lanes 16 bytes apart: conflict-free, and no
// im2col copy exists anywhere.  One v_mfma_f32_16x16x32_bf16 per (16 pixels, 16 channels, image row): K = 7 x 32 (the packed weights'
// zero 8th row of the row-pair form is not touched).  Weights ([kh][cout][32], 28 KB) stay in LDS for the lifetime of the persistent
// workgroup; two workgroups per CU cover each other's load and store phases.  Epilogue = conv3_epilogue (same row permutation).
struct StemKArgs {
  Conv3KArgs k;       // k.a: the launch; k.TH output rows per tile, k.tiles_per_img, k.tiles (WP, F, npix...: unused)
  int FR;             // 16-pixel fragments per output row (Wout / 16)
  int F;              // fragments per tile (TH * FR <= 32)
  int pitch;          // bytes of a padded input row (Win * 8)
  int rows_in;        // input rows per tile (2 TH + 5)
  unsigned img_bytes; // one padded image
};
constexpr int STEM_W_BYTES = 7 * C3 * 64;

template <int STATS>
__global__ __launch_bounds__(256, 2) void stem_direct_kernel(const StemKArgs sp) {
  const Conv3KArgs& kp = sp.k;
  const IgemmArgs& p = kp.a;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_w = smem;
  char* lds_x = smem + STEM_W_BYTES;
  const int x_bytes = sp.rows_in * sp.pitch;
  float* stat_acc = reinterpret_cast<float*>(smem + STEM_W_BYTES + 2 * ((x_bytes + 1023) & ~1023));  // [4 waves][64 channels][2]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int bx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7u) * (G >> 3) + min((int)(blockIdx.x & 7u), G & 7) + (int)(blockIdx.x >> 3));
  const int px = lane & 15, q4 = lane >> 4;

  // weights: LDS row = kh * 64 + r (64 bytes: the 8 pixels x 4 channels of image row kh), r = the permuted output channel of
  // conv3_epilogue; 16-byte chunk index XORed with (r >> 2) & 3 — the 16 rows of a fragment read then cover every bank once
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.wt);  // packed [cout][8 rows][32]
    for (int j = tid; j < 7 * C3 * 4; j += 256) {
      const int row = j >> 2, ch = j & 3;
      const int kh = row >> 6, r = row & 63;
      const int co = 32 * (r >> 5) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3);
      *reinterpret_cast<uint4*>(lds_w + row * 64 + ((ch ^ ((r >> 2) & 3)) << 4)) = wsrc[(co * 8 + kh) * 4 + ch];
    }
  }
  if constexpr (STATS != 0) {
    for (int i = tid; i < 4 * C3 * 2; i += 256) stat_acc[i] = 0.f;
  }

  // fragment mt of this wave = fragment f = mt * 4 + wave of the tile: output row f / FR of the tile, columns 16 (f % FR) ...
  constexpr int MT = 8;
  int orel[MT], abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int f = mt * 4 + wave;
    const bool ok = f < sp.F;
    const int ro = ok ? f / sp.FR : 0, fc = ok ? f - ro * sp.FR : 0;
    orel[mt] = ok ? ro * p.Wout + 16 * fc + px : -1;
    abase[mt] = 2 * ro * sp.pitch + 16 * (16 * fc + px) + 16 * q4;
  }
  // a tile's input rows are contiguous in memory: 1 KiB LDS-DMA pieces, piece pc by wave pc % 4, into one of two LDS images — the
  // NEXT tile streams in while this one is computed (no registers, no wait until after the MFMAs)
  const i32x4 srdI = make_srd(p.in, (unsigned)((size_t)p.N * sp.img_bytes));
  const int npieces = (x_bytes + 1023) >> 10;
  const unsigned xb = (unsigned)npieces << 10;  // bytes of one LDS image
  auto dma = [&](int tile, int buf) __attribute__((always_inline)) {
    const int n = tile / kp.tiles_per_img;
    const int r0 = (tile - n * kp.tiles_per_img) * kp.TH;
    const unsigned src = (unsigned)n * sp.img_bytes + (unsigned)(2 * r0) * (unsigned)sp.pitch + (unsigned)lane * 16u;
    const unsigned dst = lds_addr(lds_x) + (unsigned)buf * xb;
    for (int pc = wave; pc < npieces; pc += 4) blds16(srdI, src + ((unsigned)pc << 10), dst + ((unsigned)pc << 10));
  };
  int buf = 0;
  if (bx < kp.tiles) dma(bx, 0);
  MI355_WAIT_VM(0);
  MI355_LDS_BARRIER();  // weights (ds_write) + the first image

  for (int tile = bx; tile < kp.tiles; tile += G) {
    const int next = tile + G;
    if (next < kp.tiles) dma(next, buf ^ 1);
    const char* img = lds_x + (unsigned)buf * xb;

    f32x4 acc[MT][NT3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      bf16x8 A[MT], B[NT3];
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) {
        const int r = nt * 16 + px;
        B[nt] = *reinterpret_cast<const bf16x8*>(lds_w + (kh * C3 + r) * 64 + ((q4 ^ ((r >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) A[mt] = *reinterpret_cast<const bf16x8*>(img + abase[mt] + kh * sp.pitch);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[nt], A[mt], acc[mt][nt], 0, 0, 0);
    }
    // the next image has had the MFMA phase to land; waiting here (not behind the epilogue) keeps this tile's stores out of the wait
    MI355_WAIT_VM(0);
    conv3_epilogue<STATS, MT>(kp, acc, orel, tile, stat_acc, wave, px, q4, tid);
    MI355_LDS_BARRIER();  // every wave's pieces of the next image are in LDS, and nobody reads this one any more
    buf ^= 1;
  }

  if constexpr (STATS != 0) {
    __syncthreads();
    float* row = p.stat_partial + (size_t)bx * 2 * C3;
    for (int c = tid; c < C3; c += 256) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {  // the waves, in order
        a += stat_acc[(w * C3 + c) * 2];
        b += stat_acc[(w * C3 + c) * 2 + 1];
      }
      row[c] = a;
      row[C3 + c] = b;
    }
  }
}

// weights + two images of a tile's input rows (whole 1 KiB DMA pieces) + the statistics rows
size_t stem_lds_bytes(int th, int pitch) { return (size_t)STEM_W_BYTES + 2 * (((size_t)(2 * th + 5) * pitch + 1023) & ~(size_t)1023) + 4 * C3 * 2 * sizeof(float); }

// output rows per tile: TH | Hout, TH * FR <= 32 fragments, everything within half a CU's LDS
int plan_stem(int Hout, int Wout, int pitch) {
  if (Wout % 16) return 0;
  const int FR = Wout / 16;
  int best = 0;
  for (int th = 1; th <= Hout; ++th) {
    if (Hout % th || th * FR > 32) continue;
    if (stem_lds_bytes(th, pitch) > 80 * 1024) continue;
    best = th;
  }
  return best;
}

unsigned magic32_3(unsigned d) { return (unsigned)((1ull << 32) / d + 1); }

// tile geometry: TH | H rows, row pitch WP >= W + 2 with TH * WP a multiple of 16, at most 32 fragments, the LDS image within
// the prefetch registers (19 x 256 chunks) — the smallest padded area per output pixel wins
}  // namespace

// true when launch_stem_direct can run this launch: the bf16 stem as build_stem_fwd_args() describes it (4 row-pair taps over
// NHWC4 rows, stride 2, 64 output channels), no addend, no BN-backward sums
bool stem_direct_legal(const IgemmArgs& a, int nclass) {
  if (nclass != 1 || a.Ck != STEM_CK || a.Ncols != C3 || a.pix_stride != STEM_PS || a.IS != 2 || a.OS != 1 || a.wtaps != 4 || a.cls[0].ntaps != 4) return false;
  if (a.pair_delta != a.Win * STEM_PS - 32 || a.addend || a.bn_y) return false;
  if (a.Hsub != a.Hout || a.Wsub != a.Wout || a.Hin != 2 * a.Hout + 2 * STEM_PAD || a.Win < 2 * a.Wout + 6) return false;
  for (int t = 0; t < 4; ++t)
    if (a.cls[0].taps[t].dh != 2 * t || a.cls[0].taps[t].dw != 0 || a.cls[0].taps[t].wtap != t) return false;
  if ((unsigned long long)a.N * a.Hout * a.Wout * C3 * 2 >= 0x80000000ull) return false;
  return plan_stem(a.Hout, a.Wout, a.Win * STEM_PS * 2) > 0;
}

int launch_stem_direct(const IgemmArgs& a, hipStream_t stream, int* stat_rows) {
  MI355_ARG(stem_direct_legal(a, 1), "stem: unsupported geometry");
  StemKArgs k{};
  k.k.a = a;
  k.pitch = a.Win * STEM_PS * 2;
  k.k.TH = plan_stem(a.Hout, a.Wout, k.pitch);
  if (knobs().stem_th > 0) {  // MI355_STEM_TH, A/B knob: a smaller tile (must divide Hout)
    const int th = knobs().stem_th;
    if (th >= 1 && th <= k.k.TH && a.Hout % th == 0) k.k.TH = th;
  }
  k.FR = a.Wout / 16;
  k.F = k.k.TH * k.FR;
  k.rows_in = 2 * k.k.TH + 5;
  k.k.tiles_per_img = a.Hout / k.k.TH;
  k.k.tiles = a.N * k.k.tiles_per_img;
  k.img_bytes = (unsigned)((size_t)a.Hin * k.pitch);
  const int cus = device_cus();
  const int grid = k.k.tiles < 2 * cus ? k.k.tiles : 2 * cus;  // two persistent workgroups per CU
  const size_t lds = stem_lds_bytes(k.k.TH, k.pitch);
  MI355_ARG((size_t)a.N * k.img_bytes < 0xfffffff0ull, "stem: input beyond 32-bit offsets");
  const bool stats = a.stat_partial != nullptr;
  if (stat_rows) *stat_rows = stats ? grid : 0;
  if (knobs().stem_dbg) {
    int nb = -1;
    lds_opt_in3((const void*)stem_direct_kernel<1>);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)stem_direct_kernel<1>, 256, lds);
    fprintf(stderr, "stem_direct: TH %d F %d lds %zu grid %d resident/CU %d\n", k.k.TH, k.F, lds, grid, nb);
  }
  if (stats) {
    lds_opt_in3((const void*)stem_direct_kernel<1>);
    hipLaunchKernelGGL(stem_direct_kernel<1>, dim3(grid), dim3(256), lds, stream, k);
  } else {
    lds_opt_in3((const void*)stem_direct_kernel<0>);
    hipLaunchKernelGGL(stem_direct_kernel<0>, dim3(grid), dim3(256), lds, stream, k);
  }
  MI355_LAUNCH_CHECK();
  note_kernel("stem_direct");
  return 0;
}

}  // namespace mi355
