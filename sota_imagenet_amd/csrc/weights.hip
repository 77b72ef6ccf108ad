// weights.hip — per-step weight preparation: fp32 KRSC master -> the operand images the conv kernels read.
//   * cast copy      [Cout][taps][Cin]  in the compute dtype (bf16 path only)
//   * transposed copy [Cin][taps][Cout] in the compute dtype (the K-contiguous B operand of dgrad)
//   * stem pack      [64][7][7][3] -> [64][4 row pairs][2 x (8 px * 4 ch)] zero padded (the stem runs as 4 row-pair taps of 64)
//   all conv layers of the network are prepared by ONE launch (weight_prep_batch_kernel, device-side layer table)
// ~94 MB of reads per step for ResNet-50: HBM-trivial, and it keeps the master weights in the layout
// torch's state_dict exposes (reference interchange: train.py:101,184).
#include <hip/hip_fp8.h>

#include "common.h"

namespace mi355 {
namespace {

template <typename T, typename Tin = float>
__global__ void weight_prep_kernel(const Tin* __restrict__ w, T* __restrict__ w_cast, T* __restrict__ w_tr,
                                   int Cout, int taps, int Cin) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const size_t src = ((size_t)(co0 + r) * taps + t) * Cin + ci0 + tx;
    const float v = (float)w[src];
    tile[r][tx] = v;
    if (w_cast) w_cast[src] = (T)v;
  }
  __syncthreads();
  if (w_tr) {
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const size_t dst = ((size_t)(ci0 + r) * taps + t) * Cout + co0 + tx;
      w_tr[dst] = (T)tile[tx][r];
    }
  }
}

// all conv layers of the network in ONE launch (53 tiny launches cost more in launch latency than in bytes): block b
// finds its layer in the prefix table and does the same 32x32 tile as weight_prep_kernel
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const PrepDesc* __restrict__ table, int nlayers, const float* __restrict__ params) {
  // PREP_TILE = 64: a row of the tile is 256 bytes of fp32 in, 128 bytes of bf16 out (cast copy and transposed copy) — the 32-wide
  // tile of round 1 wrote 64-byte segments and ran at 2.4 TB/s
  constexpr int PT = PREP_TILE;
  __shared__ float tile[PT][PT + 1];
  const int b = blockIdx.x;
  int l = 0;
  while (l + 1 < nlayers && table[l + 1].tile_begin <= b) ++l;  // uniform: scalar loads
  const PrepDesc d = table[l];
  const int local = b - d.tile_begin;
  const int nci = d.Cin / PT, nco = d.Cout / PT;
  const int ci0 = (local % nci) * PT, co0 = ((local / nci) % nco) * PT, t = local / (nci * nco);
  const float* w = params + d.w_off;
  T* w_cast = reinterpret_cast<T*>(d.w_cast);
  T* w_tr = reinterpret_cast<T*>(d.w_tr);
  // fp8 step: e4m3 twins of the bf16 copies (same two layouts) under the layer's scale of this step, + the layer's amax
  unsigned char* w_q = reinterpret_cast<unsigned char*>(d.w_q);
  unsigned char* w_trq = reinterpret_cast<unsigned char*>(d.w_trq);
  const float qs = w_q ? d.q_scale[0] : 1.f;
  float amax = 0.f;
  const int tx = threadIdx.x & (PT - 1), ty = threadIdx.x / PT;
#pragma unroll 4
  for (int r = ty; r < PT; r += 256 / PT) {
    const size_t src = ((size_t)(co0 + r) * d.taps + t) * d.Cin + ci0 + tx;
    const float v = w[src];
    tile[r][tx] = v;
    if (w_cast) w_cast[src] = (T)v;
    if (w_q) {
      const float rv = (float)(T)v;
      amax = fmaxf(amax, fabsf(rv));
      w_q[src] = __hip_cvt_float_to_fp8(rv * qs, __HIP_SATFINITE, __HIP_E4M3);
    }
  }
  __syncthreads();
  if (w_tr) {
#pragma unroll 4
    for (int r = ty; r < PT; r += 256 / PT) {
      const size_t dst = ((size_t)(ci0 + r) * d.taps + t) * d.Cout + co0 + tx;
      w_tr[dst] = (T)tile[tx][r];
      if (w_trq) w_trq[dst] = __hip_cvt_float_to_fp8((float)(T)tile[tx][r] * qs, __HIP_SATFINITE, __HIP_E4M3);
    }
  }
  if (w_q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((threadIdx.x & 63) == 0 && amax > __uint_as_float(__atomic_load_n(d.q_amax, __ATOMIC_RELAXED))) atomicMax(d.q_amax, __float_as_uint(amax));
  }
}

template <typename T>
__global__ void stem_pack_kernel(const float* __restrict__ w, T* __restrict__ packed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 4 * 64) return;
  const int e = i & 31;  // element inside one image row's half of the 64-wide slab
  const int kw = e >> 2, c = e & 3;
  const int kh = 2 * ((i >> 6) & 3) + ((i >> 5) & 1);
  const int co = i >> 8;
  float v = 0.f;
  if (kh < 7 && kw < 7 && c < 3) v = w[((co * 7 + kh) * 7 + kw) * 3 + c];
  packed[i] = (T)v;
}

}  // namespace

int launch_weight_prep(int dtype, const float* w, void* w_cast, void* w_tr, int Cout, int taps, int Cin,
                       hipStream_t stream) {
  MI355_ARG(Cout % 32 == 0 && Cin % 32 == 0, "weight_prep: Cout=%d Cin=%d must be multiples of 32", Cout, Cin);
  if (!w_cast && !w_tr) return 0;
  dim3 grid(Cin / 32, Cout / 32, taps);
  if (dtype == MI355_F32)
    hipLaunchKernelGGL((weight_prep_kernel<float, float>), grid, dim3(256), 0, stream, w, (float*)w_cast, (float*)w_tr, Cout,
                       taps, Cin);
  else
    hipLaunchKernelGGL((weight_prep_kernel<bf16_t, float>), grid, dim3(256), 0, stream, w, (bf16_t*)w_cast, (bf16_t*)w_tr,
                       Cout, taps, Cin);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_transpose_any(int dtype, const void* w, void* wt, int Cout, int taps, int Cin, hipStream_t stream) {
  MI355_ARG(Cout % 32 == 0 && Cin % 32 == 0, "transpose: Cout=%d Cin=%d must be multiples of 32", Cout, Cin);
  dim3 grid(Cin / 32, Cout / 32, taps);
  if (dtype == MI355_F32)
    hipLaunchKernelGGL((weight_prep_kernel<float, float>), grid, dim3(256), 0, stream, (const float*)w,
                       (float*)nullptr, (float*)wt, Cout, taps, Cin);
  else
    hipLaunchKernelGGL((weight_prep_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)w,
                       (bf16_t*)nullptr, (bf16_t*)wt, Cout, taps, Cin);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_weight_prep_batch(int dtype, const PrepDesc* table, int nlayers, int total_tiles, const float* params,
                             hipStream_t stream) {
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(weight_prep_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, stream, table, nlayers, params);
  else
    hipLaunchKernelGGL(weight_prep_batch_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, stream, table, nlayers, params);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_stem_pack(int dtype, const float* w, void* packed, hipStream_t stream) {
  const int n = 64 * 4 * 64;
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, stream, w, (float*)packed);
  else
    hipLaunchKernelGGL(stem_pack_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, stream, w, (bf16_t*)packed);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace mi355
