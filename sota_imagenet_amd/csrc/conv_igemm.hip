// conv_igemm.hip — implicit-GEMM convolution forward / data-gradient for gfx950 (MI355X).
//
// Replaces the cuDNN conv fwd / dgrad kernels that run beneath `model(data)` and `loss.backward()` in the
// reference (call form sota_imagenet/callbacks.py:316-317; model built at train.py:64).
//
// One kernel serves forward, dgrad (stride 1 and, through four output-parity classes, stride 2) and the
// 7x7 stem (as 7 row-taps over a zero-padded NHWC4 image): see IgemmArgs in common.h.
//
//   block tile  128 (pixels) x BN (channels), BN = 128 | 64;   K step = 128 bytes of the tap's channel run
//   4 waves (2x2), each 64 x BN/2, built from 32x32 MFMA tiles:
//       fp32 : v_mfma_f32_32x32x2_f32   (exact fp32 fma chain — the parity path)
//       bf16 : v_mfma_f32_32x32x16_bf16 (fp32 accumulate)
//   A rows are gathered pixel runs (NHWC => the tap's Cin run is contiguous), staged global -> VGPR -> LDS
//   with the next tile's loads in flight during the MFMAs of the current one (register double buffer).
//   LDS rows are 128 B + 16 B pad: conflict-free for ds_write_b128 staging and ds_read_b128 fragments.
//   Because k is only a summation index, an fp32 lane fetches 4 consecutive k with one ds_read_b128 and
//   feeds them to 4 successive 32x32x2 MFMAs (A and B use the same permutation).
#include "common.h"
#include "vec.h"

namespace mi355 {

namespace {

constexpr int BM = 128;
constexpr int BKB = 128;           // bytes of K per step
constexpr int LDS_ROW = BKB + 16;  // padded LDS row, bytes

template <typename T, int BN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs p) {
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int NI = BN / 64;       // 32-col MFMA tiles per wave along N
  constexpr int NB_LD = BN / 32;    // B staging chunks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + BM * LDS_ROW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const TapClass& cls = p.cls[blockIdx.z];
  const int Msub = p.N * p.Hsub * p.Wsub;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread staging rows ------------------------------------------------------------------
  const int srow = tid >> 3;    // 0..31
  const int schunk = tid & 7;   // 16-byte chunk within the 128-byte K slab
  int a_hb[4], a_h[4], a_w[4];  // (n*Hin), i*IS, j*IS ; a_hb < 0 marks an out-of-range row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + srow + 32 * i;
    if (m < Msub) {
      int j = m % p.Wsub;
      int t = m / p.Wsub;
      int ii = t % p.Hsub;
      int n = t / p.Hsub;
      a_hb[i] = n * p.Hin;
      a_h[i] = ii * p.IS;
      a_w[i] = j * p.IS;
    } else {
      a_hb[i] = -1;
      a_h[i] = 0;
      a_w[i] = 0;
    }
  }

  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int kc_per_tap = p.Ck / BK;
  const int nk = cls.ntaps * kc_per_tap;

  uint4 ra[4], rb[NB_LD];
  const char* in_base = (const char*)p.in;
  const char* wt_base = (const char*)p.wt;

  auto load_tile = [&](int kt) {
    const int t = kt / kc_per_tap;
    const int c0 = (kt - t * kc_per_tap) * BK;
    const Tap tp = cls.taps[t];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ih = a_h[i] + tp.dh;
      const int iw = a_w[i] + tp.dw;
      const bool ok = (a_hb[i] >= 0) && ((unsigned)ih < (unsigned)p.Hin) && ((unsigned)iw < (unsigned)p.Win);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) {
        const size_t pix = (size_t)(a_hb[i] + ih) * p.Win + iw;
        const char* src = in_base + (pix * p.pix_stride + c0) * sizeof(T) + schunk * 16;
        v = *reinterpret_cast<const uint4*>(src);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB_LD; ++i) {
      const int col = n0 + srow + 32 * i;
      const char* src = wt_base + (((size_t)col * p.wtaps + tp.wtap) * p.Ck + c0) * sizeof(T) + schunk * 16;
      rb[i] = *reinterpret_cast<const uint4*>(src);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(As + (srow + 32 * i) * LDS_ROW + schunk * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < NB_LD; ++i)
      *reinterpret_cast<uint4*>(Bs + (srow + 32 * i) * LDS_ROW + schunk * 16) = rb[i];
  };

  // fragment row bases (bytes)
  const int frag_k = (lane >> 5) * 16;  // lane half -> which 16 bytes of each 32-byte k group
  int a_off[2], b_off[NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) a_off[mi] = (wm * 64 + mi * 32 + (lane & 31)) * LDS_ROW + frag_k;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_off[ni] = (wn * (BN / 2) + ni * 32 + (lane & 31)) * LDS_ROW + frag_k;

  if (nk > 0) {
    load_tile(0);
    store_tile();
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // four 32-byte k groups per 128-byte slab
      if constexpr (sizeof(T) == 4) {
        f32x4 av[2], bv[NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) av[mi] = *reinterpret_cast<const f32x4*>(As + a_off[mi] + g * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(Bs + b_off[ni] + g * 32);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[ni][e], av[mi][e], acc[mi][ni], 0, 0, 0);
      } else {
        bf16x8 av[2], bv[NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) av[mi] = *reinterpret_cast<const bf16x8*>(As + a_off[mi] + g * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = *reinterpret_cast<const bf16x8*>(Bs + b_off[ni] + g * 32);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[ni], av[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // The MFMAs ran with swapped operands (D^T = W * A^T), so a lane holds ONE pixel (lane&31) and, per group of 4
  // accumulator registers, 4 CONSECUTIVE channels: 8*g + 4*(lane>>5) + {0..3}.  Each wave stages its 32 x BN/2
  // sub-tile in fp32 through a private LDS region (ds_write_b128) and reads it back row-contiguous, so every
  // global store / addend load is 16 bytes per lane and 64..256 contiguous bytes per pixel row.
  constexpr int WN = BN / 2;                    // channels per wave
  constexpr int ST_ROW = WN * 4 + 16;           // staged fp32 row + pad (bytes)
  constexpr int VEC = 16 / (int)sizeof(T);      // output elements per 16-byte store
  constexpr int CPR = WN / VEC;                 // 16-byte chunks per staged row
  constexpr int RPI = 64 / CPR;                 // rows covered by one wave-instruction
  int* row_pix = reinterpret_cast<int*>(smem + 4 * 32 * ST_ROW);
  char* stage = smem + wave * 32 * ST_ROW;
  if (tid < BM) {
    int m = m0 + tid;
    int pix = -1;
    if (m < Msub) {
      int j = m % p.Wsub;
      int t = m / p.Wsub;
      int ii = t % p.Hsub;
      int n = t / p.Hsub;
      pix = (n * p.Hout + ii * p.OS + cls.ph) * p.Wout + j * p.OS + cls.pw;
    }
    row_pix[tid] = pix;
  }

  T* out = reinterpret_cast<T*>(p.out);
  const T* addend = reinterpret_cast<const T*>(p.addend);
  const int prow = lane & 31, hh = lane >> 5;
  const int rr = lane / CPR, ch = lane % CPR;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    __syncthreads();  // previous pass fully read (and, first time, row_pix written / main loop done with LDS)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
        *reinterpret_cast<f32x4*>(stage + prow * ST_ROW + (ni * 32 + 8 * g + 4 * hh) * 4) = v;
      }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 32 / RPI; ++ps) {
      const int row = ps * RPI + rr;
      const int pix = row_pix[wm * 64 + mi * 32 + row];
      if (pix < 0) continue;
      const size_t o = (size_t)pix * p.Ncols + n0 + wn * WN + ch * VEC;
      float v[VEC];
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + row * ST_ROW + (ch * VEC + 4 * q) * 4);
        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
      }
      if (addend) {
        float a[VEC];
        Vec16<T>::load(addend + o, a);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += a[e];
      }
      Vec16<T>::store(out + o, v);
    }
  }
}

template <typename T, int BN>
int launch_t(const IgemmArgs& a, int nclass, hipStream_t stream) {
  const int Msub = a.N * a.Hsub * a.Wsub;
  dim3 grid(cdiv(Msub, BM), a.Ncols / BN, nclass);
  size_t lds = (size_t)(BM + BN) * LDS_ROW;
  const size_t lds_epi = (size_t)4 * 32 * ((BN / 2) * 4 + 16) + BM * sizeof(int);
  if (lds_epi > lds) lds = lds_epi;
  hipLaunchKernelGGL((igemm_kernel<T, BN>), grid, dim3(256), lds, stream, a);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int launch_igemm(int dtype, const IgemmArgs& a, int nclass, hipStream_t stream) {
  const int bk = BKB / (int)dtype_size(dtype);
  MI355_ARG(a.in && a.wt && a.out, "igemm: null pointer");
  MI355_ARG(a.Ck % bk == 0, "igemm: Ck=%d not a multiple of %d", a.Ck, bk);
  MI355_ARG(a.Ncols % 64 == 0, "igemm: Ncols=%d not a multiple of 64", a.Ncols);
  MI355_ARG(nclass >= 1 && nclass <= 4, "igemm: nclass=%d", nclass);
  MI355_ARG(((size_t)a.pix_stride * dtype_size(dtype)) % 8 == 0, "igemm: pixel stride not 8-byte aligned");
  const bool wide = (a.Ncols % 128 == 0);
  if (dtype == MI355_F32) return wide ? launch_t<float, 128>(a, nclass, stream) : launch_t<float, 64>(a, nclass, stream);
  if (dtype == MI355_BF16)
    return wide ? launch_t<bf16_t, 128>(a, nclass, stream) : launch_t<bf16_t, 64>(a, nclass, stream);
  set_error("igemm: bad dtype %d", dtype);
  return MI355_E_ARG;
}

}  // namespace mi355
