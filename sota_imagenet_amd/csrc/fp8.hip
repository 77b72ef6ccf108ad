// fp8.hip — OCP fp8 (e4m3fn) operand path of the convolution for gfx950 (MI355X): quantisation + the per-op entry points.
//
// BASELINE.json configs[4] asks for "fp8 MFMA convs".  gfx950 runs the non-scaled fp8 MFMAs at the bf16 matrix rate
// (MI355X_MICROARCH.md § Matrix cores), so what fp8 operands buy is bytes: half the HBM / L2 / LDS traffic per MAC, and —
// decisive for the 8-wave kernel, which is bound by its fragment reads — one ds_read_b128 per TWO MFMAs.
// The convolution itself is conv_igemm8.hip with EB = 1; this file holds the element-wise quantiser and the C-ABI wrappers.
// Scaling is per tensor: q = sat_e4m3(x * scale); the conv output is (sum q_x * q_w) / (scale_x * scale_w) in bf16.
#include <hip/hip_fp8.h>

#include "common.h"
#include "vec.h"

namespace mi355 {

int launch_igemm8_fp8(const IgemmArgs& a, int nclass, int bm, int bn, int korder, float oscale, hipStream_t stream, int* stat_rows);
bool igemm8_fp8_legal(const IgemmArgs& a, int nclass, int bn);

namespace {

template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* x, unsigned char* q, float scale, size_t nvec) {
  constexpr int V = Vec16<T>::N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float v[V];
    Vec16<T>::load(x + i * V, v);
    unsigned char o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = __hip_cvt_float_to_fp8(v[e] * scale, __HIP_SATFINITE, __HIP_E4M3);
    if constexpr (V == 8) {
      *reinterpret_cast<uint2*>(q + i * V) = *reinterpret_cast<const uint2*>(o);
    } else {
      *reinterpret_cast<unsigned*>(q + i * V) = *reinterpret_cast<const unsigned*>(o);
    }
  }
}

int pick_tile(const IgemmArgs& a, int nclass, int* bm, int* bn) {
  // the wide tile when it alone fills most of the 256 CUs (same threshold as the bf16 rule, conv_igemm.hip::choose_igemm8)
  const long long M = (long long)a.N * a.Hsub * a.Wsub;
  if (a.Ncols % 256 == 0 && ((M + 223) / 224) * nclass * (a.Ncols / 256) * 10 >= 7LL * device_cus()) { *bm = 224; *bn = 256; return 0; }
  if (a.Ncols % 128 == 0) { *bm = 256; *bn = 128; return 0; }
  set_error("conv fp8: %d output columns (a multiple of 128 is needed)", a.Ncols);
  return MI355_E_ARG;
}

// delayed per-tensor scaling: the amax a tensor's producer recorded in the previous step sets this step's scale
__global__ void fp8_scale_update_kernel(float* scale, unsigned* amax, int n, float headroom) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = __uint_as_float(amax[i]);
  if (a > 0.f && a < 3.0e38f) scale[i] = 448.f / (headroom * a);
  amax[i] = 0u;
}

}  // namespace

bool igemm_fp8_legal(const IgemmArgs& a, int nclass) { return igemm8_fp8_legal(a, nclass, 128); }

int launch_igemm_fp8(const IgemmArgs& a, int nclass, float oscale, hipStream_t stream, int* stat_rows) {
  // the stride-1 3x3 launches of layers 2 - 4: the generated direct kernel on the K = 128 MFMA (asm/dconv_gen.py Cfg.fp8)
  if (dconv_fp8_legal(a, nclass)) return launch_dconv_fp8(a, nclass, oscale, stream, stat_rows);
  int bm, bn;
  MI355_TRY(pick_tile(a, nclass, &bm, &bn));
  int max_taps = 0;
  for (int ci = 0; ci < nclass; ++ci) max_taps = a.cls[ci].ntaps > max_taps ? a.cls[ci].ntaps : max_taps;
  return launch_igemm8_fp8(a, nclass, bm, bn, max_taps > 1 ? 1 : 0, oscale, stream, stat_rows);
}

int launch_fp8_scale_update(float* scale, unsigned* amax, int n, float headroom, hipStream_t s) {
  hipLaunchKernelGGL(fp8_scale_update_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, scale, amax, n, headroom);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_quantize_fp8(int src_dtype, const void* x, void* q, float scale, size_t n, void* stream) {
  MI355_ARG(x && q && n % 8 == 0 && scale > 0.f, "quantize_fp8: n=%zu must be a multiple of 8, scale > 0", n);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == MI355_F32) {
    const size_t nvec = n / 4;
    hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3((unsigned)std::min<size_t>((nvec + 255) / 256, 2048)), dim3(256), 0, s, (const float*)x, (unsigned char*)q, scale, nvec);
  } else if (src_dtype == MI355_BF16) {
    const size_t nvec = n / 8;
    hipLaunchKernelGGL(quantize_fp8_kernel<bf16_t>, dim3((unsigned)std::min<size_t>((nvec + 255) / 256, 2048)), dim3(256), 0, s, (const bf16_t*)x, (unsigned char*)q, scale, nvec);
  } else {
    set_error("quantize_fp8: bad source dtype %d", src_dtype);
    return MI355_E_ARG;
  }
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

__global__ void set_two_scalars(float* p, float a, float b) {
  if (threadIdx.x == 0) { p[0] = a; p[1] = b; }
}

extern "C" {

int mi355_conv2d_fwd_fp8(const void* xq, const void* wq, void* y, float oscale, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                         int pad, void* stream) {
  MI355_ARG(xq && wq && y && Cin % 128 == 0 && Cout % 128 == 0 && KH * KW <= 9 && (stride == 1 || stride == 2), "conv2d_fwd_fp8: Cin=%d Cout=%d (multiples of 128)", Cin, Cout);
  IgemmArgs a;
  build_fwd_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  a.in = xq; a.wt = wq; a.out = y;
  return launch_igemm_fp8(a, 1, oscale, (hipStream_t)stream);
}

int mi355_conv2d_wgrad_fp8(const void* dyq, const void* xq, float* dw, float beta, float oscale, int N, int H, int W, int Cin, int Cout,
                           int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  MI355_ARG(dyq && xq && dw && Cin % 128 == 0 && Cout % 128 == 0 && KH * KW <= 9 && (stride == 1 || stride == 2),
            "conv2d_wgrad_fp8: Cin=%d Cout=%d (multiples of 128)", Cin, Cout);
  WgradArgs a;
  build_wgrad_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  const int splits = plan_wgrad_splits(MI355_BF16, a.N * a.Ho * a.Wo, Cout, KH * KW, Cin);
  const size_t n = (size_t)Cout * KH * KW * Cin;
  MI355_ARG(ws && ws_bytes >= (size_t)splits * n * 4 + 256, "wgrad fp8: workspace too small (%zu < %zu)", ws_bytes, (size_t)splits * n * 4 + 256);
  a.dy = dyq; a.x = xq; a.partial = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(launch_wgrad(MI355_FP8, a, splits, s));
  // the per-op form takes host scales: park 1 / oscale and 1 behind the partial slabs as the two device scalars the reduce reads
  float* sc = (float*)((char*)ws + (size_t)splits * n * 4);
  hipLaunchKernelGGL(set_two_scalars, dim3(1), dim3(64), 0, s, sc, 1.f / oscale, 1.f);  // (values travel as kernel arguments: no host buffer outlives the call)
  MI355_HIP(hipGetLastError());
  return launch_splitk_reduce((const float*)ws, splits, n, dw, n, beta, s, sc, sc + 1);
}

int mi355_conv2d_dgrad_fp8(const void* dyq, const void* wtq, void* dx, float oscale, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, void* stream) {
  MI355_ARG(dyq && wtq && dx && Cin % 128 == 0 && Cout % 128 == 0 && KH * KW <= 9, "conv2d_dgrad_fp8: Cin=%d Cout=%d (multiples of 128)", Cin, Cout);
  IgemmArgs a;
  const int nclass = build_dgrad_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  if (nclass < 0) return nclass;
  a.in = dyq; a.wt = wtq; a.out = dx;
  return launch_igemm_fp8(a, nclass, oscale, (hipStream_t)stream);
}

}  // extern "C"
