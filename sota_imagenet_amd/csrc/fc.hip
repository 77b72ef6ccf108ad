// fc.hip — the classifier head's two small fp32 GEMMs for gfx950 (MI355X):  out[m][n] = sum_k x[m][k] * w[n][k].
//
// Replaces the ATen linear forward / input-gradient under `model(data)` / `loss.backward()` (sota_imagenet/callbacks.py:316-317;
// pytorch_tools resnet50's `last_linear`).  M = batch (256), K x N = 2048 x 1000 forward (weights [N][K]) and 1024 x 2048 for
// the gradient wrt the pooled features (transposed weights).  As a 1x1 convolution this is 32 (64) tiles of the conv kernel on
// 256 CUs with a 2048-long reduction each: 103 us + 53 us per step, both on the critical path.  Here every workgroup owns a
// 32 x 64 output tile and its four waves split K four ways (128 resp. 256 workgroups, one wave per SIMD):
//   operands  straight from global memory (L2-resident: 8 MB of weights), 16 bytes per lane = 4 consecutive k of one row, which
//             feed 4 x v_mfma_f32_32x32x2_f32 — lane half h takes k = 8j + 4h + {0..3} on BOTH operands, so the k-to-lane
//             assignment of the instruction need not be known (k is a summation index); exact fp32 FMAs, fp32 accumulate
//   reduce    the four waves' partial tiles meet in LDS and are added in wave order (fixed => bitwise reproducible)
#include "common.h"

namespace mi355 {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int M, int N, int K) {
  __shared__ float part[4][32][64 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 32;
  const int kper = K / 4, kbeg = wave * kper;
  const int m = m0 + r;
  const float* xr = x + (size_t)(m < M ? m : 0) * K + kbeg + 4 * h;
  const float* w0 = w + (size_t)(n0 + r) * K + kbeg + 4 * h;
  const float* w1 = w0 + (size_t)32 * K;
  f32x16 acc0 = {0}, acc1 = {0};
  const bool live = m < M;
  for (int k = 0; k < kper; k += 16) {  // two 8-wide k groups per trip: six 16-byte loads in flight per lane
    f32x4 a[2], b0[2], b1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      a[u] = *reinterpret_cast<const f32x4*>(xr + k + 8 * u);
      b0[u] = *reinterpret_cast<const f32x4*>(w0 + k + 8 * u);
      b1[u] = *reinterpret_cast<const f32x4*>(w1 + k + 8 * u);
      if (!live) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[u][e], a[u][e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[u][e], a[u][e], acc1, 0, 0, 0);
      }
  }
  // D[i = weight row][j = x row]: lane (r, h) holds column j = r, rows 8g + 4h + e (g, e = 0..3)
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      part[wave][r][8 * g + 4 * h + e] = acc0[4 * g + e];
      part[wave][r][32 + 8 * g + 4 * h + e] = acc1[4 * g + e];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 16; i += 256) {
    const int row = i >> 4, c4 = (i & 15) * 4;
    if (m0 + row >= M) continue;
    f32x4 s = *reinterpret_cast<const f32x4*>(&part[0][row][c4]);
#pragma unroll
    for (int v = 1; v < 4; ++v) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&part[v][row][c4]);
      s += t;
    }
    *reinterpret_cast<f32x4*>(out + (size_t)(m0 + row) * N + n0 + c4) = s;
  }
}

}  // namespace

// out[M][N] = x[M][K] * w[N][K]^T, all fp32 row-major; N % 64 == 0, K % 64 == 0
int launch_fc(const float* x, const float* w, float* out, int M, int N, int K, hipStream_t s) {
  MI355_ARG(x && w && out && M > 0 && N % 64 == 0 && K % 64 == 0, "fc: M=%d N=%d (multiple of 64) K=%d (multiple of 64)", M, N, K);
  hipLaunchKernelGGL(fc_kernel, dim3((unsigned)(N / 64), (unsigned)cdiv(M, 32)), dim3(256), 0, s, x, w, out, M, N, K);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace mi355
