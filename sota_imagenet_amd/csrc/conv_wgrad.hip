// conv_wgrad.hip — convolution weight-gradient for gfx950 (MI355X): split-K MFMA GEMM + deterministic reduce.
//
// Replaces the cuDNN wgrad kernels under `loss.backward()` in the reference (sota_imagenet/callbacks.py:317).
//
//   dW[co][tap][ci] = sum over output pixels m of dy[m][co] * x[src(m,tap)][ci]
// is a GEMM whose reduction index (the pixel) is the SLOW index of both operands in NHWC memory, so both
// MFMA operands are "transposed".  Tiles are staged in their natural [pixel][channel] layout
// (coalesced 16-byte loads, zero-filled halo) and read transposed out of LDS:
//   fp32 : ds_read_b32 (a lane holds ONE k per v_mfma_f32_32x32x2_f32 operand, lanes run along channels)
//   bf16 : ds_read_b64_tr_b16 x2 per v_mfma_f32_32x32x16_bf16 operand (row stride == 16 dwords mod 64
//          => the 4 k-rows x 32 columns of a half-wave cover all 64 banks exactly once)
// Block tile: BMC (128|64) output channels x 64 input channels of ONE tap; the pixel range is split
// over blockIdx.y; fp32 partial slabs are summed in split order by splitk_reduce (bitwise reproducible,
// no float atomics).
#include "common.h"

namespace mi355 {

namespace {

constexpr int BNC = 64;

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
  int M;
};

template <typename T, int BMC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradKArgs kp) {
  const WgradArgs& p = kp.a;
  constexpr int ES = (int)sizeof(T);
  constexpr int BKP = 128 / ES;                 // pixels per chunk: 32 fp32, 64 bf16
  constexpr int SA = BMC * ES + 64;             // LDS row strides (bytes), == 16 dwords (mod 64) for bf16
  constexpr int SB = BNC * ES + 64;
  constexpr int CPR_A = BMC * ES / 16;          // 16-byte chunks per dy row
  constexpr int CPR_B = BNC * ES / 16;
  constexpr int NLD_A = BKP * CPR_A / 256;
  constexpr int NLD_B = BKP * CPR_B / 256;
  constexpr int MI = BMC / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ad = smem;
  char* Bx = smem + BKP * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // tile decode: x = ((cb * ntaps + t) * ckb + cib)
  const int ckb = p.Ck / BNC;
  int tile = blockIdx.x;
  const int cib = tile % ckb;
  tile /= ckb;
  const int t = tile % p.ntaps;
  const int cb = tile / p.ntaps;
  const Tap tp = p.taps[t];
  const int co0 = cb * BMC;
  const int ci0 = cib * BNC;
  const int split = blockIdx.y;

  const int chunk0 = split * p.chunks_per_split;
  const int mbeg = chunk0 * BKP;
  int mend = mbeg + p.chunks_per_split * BKP;
  if (mend > kp.M) mend = kp.M;

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

  uint4 ra[NLD_A], rb[NLD_B];
  const char* dy_base = (const char*)p.dy;
  const char* x_base = (const char*)p.x;

  auto load_chunk = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NLD_A; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR_A, ch = idx % CPR_A;
      const int m = m0 + row;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (m < mend) v = *reinterpret_cast<const uint4*>(dy_base + ((size_t)m * p.Cout + co0) * ES + ch * 16);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NLD_B; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR_B, ch = idx % CPR_B;
      const int m = m0 + row;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (m < mend) {
        const uint32_t q1 = fdiv((uint32_t)m, kp.dWo);
        const int ow = m - (int)q1 * p.Wo;
        const uint32_t n = fdiv(q1, kp.dHo);
        const int oh = (int)q1 - (int)n * p.Ho;
        const int ih = oh * p.IS + tp.dh;
        const int iw = ow * p.IS + tp.dw;
        if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) {
          const size_t pix = ((size_t)n * p.Hin + ih) * p.Win + iw;
          v = *reinterpret_cast<const uint4*>(x_base + (pix * p.pix_stride + ci0) * ES + ch * 16);
        }
      }
      rb[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NLD_A; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<uint4*>(Ad + (idx / CPR_A) * SA + (idx % CPR_A) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NLD_B; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<uint4*>(Bx + (idx / CPR_B) * SB + (idx % CPR_B) * 16) = rb[i];
    }
  };

  const int ibase = wm * (BMC / 2);  // + mi*32
  const int jbase = wn * 32;

  if (mbeg < mend) {
    load_chunk(mbeg);
    store_chunk();
  }
  __syncthreads();

  for (int m0 = mbeg; m0 < mend; m0 += BKP) {
    const bool more = (m0 + BKP) < mend;
    if (more) load_chunk(m0 + BKP);
    if constexpr (ES == 4) {
      const int h = lane >> 5, c = lane & 31;
#pragma unroll
      for (int kk = 0; kk < BKP / 2; ++kk) {
        const int k = 2 * kk + h;
        const float b = *reinterpret_cast<const float*>(Bx + k * SB + (jbase + c) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float a = *reinterpret_cast<const float*>(Ad + k * SA + (ibase + mi * 32 + c) * 4);
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi], 0, 0, 0);
        }
      }
    } else {
      const int g = lane >> 4, w = lane & 15, q = w >> 2, pp = w & 3;
      const int krow = 8 * (g >> 1) + q;
      const int coff = 16 * (g & 1) + 4 * pp;
#pragma unroll
      for (int ks = 0; ks < BKP / 16; ++ks) {
        const int k0 = ks * 16 + krow;
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
        const char* bp = Bx + k0 * SB + (jbase + coff) * 2;
        s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp));
        s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp + 4 * SB));
        bf16x8 bfrag;
        {
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 tmp = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
          bfrag = __builtin_bit_cast(bf16x8, tmp);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const char* ap = Ad + k0 * SA + (ibase + mi * 32 + coff) * 2;
          s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
          s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * SA));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 tmp = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
          bf16x8 afrag = __builtin_bit_cast(bf16x8, tmp);
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[mi], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // partial[split][co][wtap][ck]
  float* outp = p.partial + (size_t)split * p.Cout * p.wtaps * p.Ck;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + ibase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ci = ci0 + jbase + (lane & 31);
      outp[((size_t)co * p.wtaps + tp.wtap) * p.Ck + ci] = acc[mi][r];
    }
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, size_t stride,
                                     float* __restrict__ dst, size_t n4, float beta) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * stride + i * 4);
    f32x4* d = reinterpret_cast<f32x4*>(dst + i * 4);
    if (beta != 0.f) s += beta * (*d);
    *d = s;
  }
}

// dw[64][7][7][3] <- partial[s][64][7][64] (kw*4 + c)
__global__ void stem_unpack_kernel(const float* __restrict__ partial, int splits, float* __restrict__ dw, float beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 7 * 7 * 3) return;
  const int c = i % 3;
  int t = i / 3;
  const int kw = t % 7;
  t /= 7;
  const int kh = t % 7;
  const int co = t / 7;
  const size_t src = ((size_t)co * 7 + kh) * 64 + kw * 4 + c;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(size_t)k * 64 * 7 * 64 + src];
  dw[i] = (beta != 0.f ? beta * dw[i] : 0.f) + s;
}

template <typename T, int BMC>
int launch_t(const WgradArgs& a, int splits, hipStream_t stream) {
  constexpr int ES = (int)sizeof(T);
  constexpr int BKP = 128 / ES;
  WgradKArgs k;
  k.a = a;
  k.M = a.N * a.Ho * a.Wo;
  k.dWo = make_fastdiv((uint32_t)a.Wo);
  k.dHo = make_fastdiv((uint32_t)a.Ho);
  const int tiles = (a.Cout / BMC) * a.ntaps * (a.Ck / BNC);
  const size_t lds = (size_t)BKP * ((BMC * ES + 64) + (BNC * ES + 64));
  hipLaunchKernelGGL((wgrad_kernel<T, BMC>), dim3(tiles, splits), dim3(256), lds, stream, k);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int plan_wgrad_splits(int M, int Cout, int ntaps, int Ck) {
  const int bmc = (Cout % 128 == 0) ? 128 : 64;
  const int tiles = (Cout / bmc) * ntaps * (Ck / BNC);
  // chunk granularity must hold for both dtypes: plan in 64-pixel units
  const int chunks = cdiv(M, 64);
  int splits = cdiv(1024, tiles);
  if (splits > chunks) splits = chunks;
  if (splits < 1) splits = 1;
  const int cps = cdiv(chunks, splits);
  return cdiv(chunks, cps);
}

int launch_wgrad(int dtype, const WgradArgs& a0, int splits, hipStream_t stream) {
  MI355_ARG(a0.dy && a0.x && a0.partial, "wgrad: null pointer");
  MI355_ARG(a0.Cout % 64 == 0 && a0.Ck % 64 == 0, "wgrad: Cout=%d Ck=%d must be multiples of 64", a0.Cout, a0.Ck);
  MI355_ARG(splits >= 1, "wgrad: splits=%d", splits);
  WgradArgs a = a0;
  const int M = a.N * a.Ho * a.Wo;
  const int bkp = 128 / (int)dtype_size(dtype);
  // splits were planned in 64-pixel units; convert to this dtype's chunk size
  const int cps64 = cdiv(cdiv(M, 64), splits);
  a.chunks_per_split = cps64 * (64 / bkp);
  const bool wide = (a.Cout % 128 == 0);
  if (dtype == MI355_F32) return wide ? launch_t<float, 128>(a, splits, stream) : launch_t<float, 64>(a, splits, stream);
  if (dtype == MI355_BF16)
    return wide ? launch_t<bf16_t, 128>(a, splits, stream) : launch_t<bf16_t, 64>(a, splits, stream);
  set_error("wgrad: bad dtype %d", dtype);
  return MI355_E_ARG;
}

int launch_splitk_reduce(const float* partial, int splits, size_t stride, float* dst, size_t n, float beta,
                         hipStream_t stream) {
  MI355_ARG(n % 4 == 0 && stride % 4 == 0, "splitk_reduce: n=%zu stride=%zu must be multiples of 4", n, stride);
  const size_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, splits, stride, dst, n4, beta);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_stem_unpack(const float* partial, int splits, float* dw, float beta, hipStream_t stream) {
  const int n = 64 * 7 * 7 * 3;
  hipLaunchKernelGGL(stem_unpack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, partial, splits, dw, beta);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace mi355
