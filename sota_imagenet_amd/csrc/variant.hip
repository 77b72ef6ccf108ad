// variant.hip — the small kernels of the BResNet-50 variant blocks (BASELINE configs[3]) for gfx950 (MI355X).
//
// The reference builds that model with `pytorch_tools.models.resnet50(stem_type="deep", antialias=True, attn_type="eca",
// norm_layer="inplaceabn", norm_act="leaky_relu", drop_rate=0.2, drop_connect_rate=0.2)`
// (configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51) and wraps every conv in weight standardisation
// (train.py:66-67).  The convolutions / BN statistics reuse the kernels of the baseline path; this file adds what is new:
//   blur pool      3x3 binomial [1,2,1] x [1,2,1] / 16, stride 2, reflect padding (anti-aliased down-sampling)
//   avg pool 2x2   the anti-aliased shortcut of a stride-2 block
//   max pool 3x3/1 the anti-aliased stem pool (max, then blur)
//   ECA            global average pool -> 1-D conv of width k over the channel axis -> sigmoid -> channel-wise scale
//   weight std     per output channel (w - mean) * rsqrt(var + eps) and its backward
//   residual + act out = act(branch * keep_scale[n] + shortcut): drop-connect scale, shortcut add, (leaky) ReLU
//   dropout        on the pooled features
// All tensors NHWC; one thread handles one 16-byte channel vector (HBM-bound streaming, coalesced over channels).
#include "common.h"
#include "vec.h"

namespace mi355 {
namespace {

constexpr float LEAKY = 0.01f;
__device__ __forceinline__ float act_fwd(float v, int act) { return act == 0 ? v : (v > 0.f ? v : (act == 2 ? v * LEAKY : 0.f)); }
__device__ __forceinline__ float act_slope(float out, int act) { return act == 0 ? 1.f : (out > 0.f ? 1.f : (act == 2 ? LEAKY : 0.f)); }

#define GRID_STRIDE(i, total) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

// pixels per thread of the row-walking pools (profiles/r05b_ab_bresnet50_steps.txt: the row-walking form against the point-wise kernels)
static int pool_seg(int W) { return 112 < W ? 112 : W; }
// workgroups of the grid-stride elementwise kernels: one per 256 elements-vectors up to a cap.  The cap was 2048 (eight
// resident rounds of one workgroup per CU) until late round 5: 32768+ is 0.3-0.4 ms per BResNet-50 step faster (28.65 -> 28.35) — short workgroups interleave
// with the other stream's kernels and leave no tail
static int grid_for(size_t total) {
  return (int)std::min<size_t>((total + 255) / 256, (size_t)65536);
}

// ---- blur pool -----------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void blurpool_fwd_kernel(const T* x, T* y, int N, int H, int W, int C) {
  constexpr int V = Vec16<T>::N;
  const int Ho = H / 2, Wo = W / 2, CV = C / V;
  const size_t total = (size_t)N * Ho * Wo * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int ih = 2 * oh - 1 + a;
      ih = ih < 0 ? -ih : ih;  // reflect (H even: the bottom edge is never crossed)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        int iw = 2 * ow - 1 + b;
        iw = iw < 0 ? -iw : iw;
        const float f = (float)((a == 1 ? 2 : 1) * (b == 1 ? 2 : 1)) * (1.f / 16.f);
        float v[V];
        Vec16<T>::load(x + (((size_t)n * H + ih) * W + iw) * C + cv * V, v);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += f * v[e];
      }
    }
    Vec16<T>::store(y + i * V, acc);
  }
}
// gather form: dx[h][w] = sum over the (output, tap) pairs that read it — taps (oh, a) with 2*oh - 1 + a == h, and for
// h == 1 additionally (oh = 0, a = 0), which reflects onto row 1
template <typename T>
__global__ __launch_bounds__(256) void blurpool_bwd_kernel(const T* dy, T* dx, int N, int H, int W, int C) {
  constexpr int V = Vec16<T>::N;
  const int Ho = H / 2, Wo = W / 2, CV = C / V;
  const size_t total = (size_t)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    // candidate (output index, filter weight) pairs along one axis
    int ohs[3], ows[3];
    float fh[3], fw[3];
    int nh = 0, nw = 0;
    for (int a = 0; a < 3; ++a) {
      const int q = h + 1 - a;  // 2*oh
      if (q >= 0 && (q & 1) == 0 && q / 2 < Ho) { ohs[nh] = q / 2; fh[nh++] = a == 1 ? 2.f : 1.f; }
    }
    if (h == 1) {  // tap a = 0 of oh = 0 reads row -1 -> 1
      bool merged = false;
      for (int k = 0; k < nh; ++k)
        if (ohs[k] == 0) { fh[k] += 1.f; merged = true; }
      if (!merged) { ohs[nh] = 0; fh[nh++] = 1.f; }
    }
    for (int b = 0; b < 3; ++b) {
      const int q = w + 1 - b;
      if (q >= 0 && (q & 1) == 0 && q / 2 < Wo) { ows[nw] = q / 2; fw[nw++] = b == 1 ? 2.f : 1.f; }
    }
    if (w == 1) {
      bool merged = false;
      for (int k = 0; k < nw; ++k)
        if (ows[k] == 0) { fw[k] += 1.f; merged = true; }
      if (!merged) { ows[nw] = 0; fw[nw++] = 1.f; }
    }
    for (int a = 0; a < nh; ++a)
      for (int b = 0; b < nw; ++b) {
        float v[V];
        Vec16<T>::load(dy + (((size_t)n * Ho + ohs[a]) * Wo + ows[b]) * C + cv * V, v);
        const float f = fh[a] * fw[b] * (1.f / 16.f);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += f * v[e];
      }
    Vec16<T>::store(dx + i * V, acc);
  }
}

// ---- 2x2 average pool ---------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void avgpool2_kernel(const T* in, T* out, int N, int H, int W, int C) {
  constexpr int V = Vec16<T>::N;
  const int Ho = H / 2, Wo = W / 2, CV = C / V;
  const size_t total = (size_t)N * (BWD ? H * W : Ho * Wo) * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    float acc[V];
    if (BWD) {
      const int w = (int)(t % W); t /= W;
      const int h = (int)(t % H);
      const int n = (int)(t / H);
      Vec16<T>::load(in + (((size_t)n * Ho + h / 2) * Wo + w / 2) * C + cv * V, acc);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] *= 0.25f;
    } else {
      const int ow = (int)(t % Wo); t /= Wo;
      const int oh = (int)(t % Ho);
      const int n = (int)(t / Ho);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float v[V];
          Vec16<T>::load(in + (((size_t)n * H + 2 * oh + a) * W + 2 * ow + b) * C + cv * V, v);
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] += 0.25f * v[e];
        }
    }
    Vec16<T>::store(out + i * V, acc);
  }
}

// ---- 3x3 stride-1 max pool (pad 1): u8 index of the first maximum in window scan order --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s1_fwd_kernel(const T* x, T* y, uint8_t* idx, int N, int H, int W, int C) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const size_t total = (size_t)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float best[V];
    int bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { best[e] = -3.0e38f; bi[e] = 4; }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ih = h - 1 + a;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int iw = w - 1 + b;
        if (iw < 0 || iw >= W) continue;
        float v[V];
        Vec16<T>::load(x + (((size_t)n * H + ih) * W + iw) * C + cv * V, v);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (v[e] > best[e]) { best[e] = v[e]; bi[e] = a * 3 + b; }
      }
    }
    Vec16<T>::store(y + i * V, best);
    // the V argmax codes of a thread are V consecutive bytes: one 4- / 8-byte store instead of V byte stores
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) w0 |= (uint32_t)bi[e] << (8 * e);
    if constexpr (V == 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) w1 |= (uint32_t)bi[4 + e] << (8 * e);
      *reinterpret_cast<uint2*>(idx + i * V) = make_uint2(w0, w1);
    } else {
      *reinterpret_cast<uint32_t*>(idx + i * V) = w0;
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s1_bwd_kernel(const T* dy, const uint8_t* idx, T* dx, int N, int H, int W, int C) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const size_t total = (size_t)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // the window centred at (h + 1 - a, w + 1 - b) reads this pixel through tap (a, b)
      const int oh = h + 1 - a;
      if (oh < 0 || oh >= H) continue;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ow = w + 1 - b;
        if (ow < 0 || ow >= W) continue;
        const size_t o = (((size_t)n * H + oh) * W + ow) * C + cv * V;
        float v[V];
        Vec16<T>::load(dy + o, v);
        uint32_t w[V / 4];
        if constexpr (V == 8) {
          const uint2 t2 = *reinterpret_cast<const uint2*>(idx + o);
          w[0] = t2.x;
          w[1] = t2.y;
        } else {
          w[0] = *reinterpret_cast<const uint32_t*>(idx + o);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += ((w[e >> 2] >> (8 * (e & 3))) & 0xffu) == (unsigned)(a * 3 + b) ? v[e] : 0.f;
      }
    }
    Vec16<T>::store(dx + i * V, acc);
  }
}

// The same two operators with one thread per (image row, channel vector) WALKING along the row: the 3 x 3 window lives in registers as three columns
// that rotate, so a pixel costs 3 vector loads instead of 9 (the point-wise forms above are bound by L2 traffic: 9 x 411 MB at the stem's 112 x 112 x 64).
// Same window scan order, same first-maximum rule, same order of the backward's additions: bit-identical results.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s1_fwd_rows_kernel(const T* x, T* y, uint8_t* idx, int N, int H, int W, int C, int SEG) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const int nseg = (W + SEG - 1) / SEG;   // a thread walks one segment of SEG pixels of a row (more threads in flight: the walk is a chain of load latencies)
  const size_t rows = (size_t)N * H * nseg * CV;
  GRID_STRIDE(i, rows) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int sg = (int)(t % nseg); t /= nseg;
    const int h = (int)(t % H), n = (int)(t / H);
    const int wb = sg * SEG, we = min(W, wb + SEG);
    float col[3][3][V];
    auto load_col = [&](float (&c)[3][V], int iw) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int ih = h - 1 + a;
        if (iw >= 0 && iw < W && ih >= 0 && ih < H) {
          Vec16<T>::load(x + (((size_t)n * H + ih) * W + iw) * C + cv * V, c[a]);
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) c[a][e] = -3.0e38f;  // (never greater than the running best: a skipped position)
        }
      }
    };
    auto emit = [&](const float (&c0)[3][V], const float (&c1)[3][V], const float (&c2)[3][V], int w) __attribute__((always_inline)) {
      float best[V];
      int bi[V];
#pragma unroll
      for (int e = 0; e < V; ++e) { best[e] = -3.0e38f; bi[e] = 4; }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          if (c0[a][e] > best[e]) { best[e] = c0[a][e]; bi[e] = a * 3; }
          if (c1[a][e] > best[e]) { best[e] = c1[a][e]; bi[e] = a * 3 + 1; }
          if (c2[a][e] > best[e]) { best[e] = c2[a][e]; bi[e] = a * 3 + 2; }
        }
      }
      const size_t o = ((((size_t)n * H + h) * W + w) * CV + cv) * V;
      Vec16<T>::store(y + o, best);
      uint32_t w0 = 0, w1 = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) w0 |= (uint32_t)bi[e] << (8 * e);
      if constexpr (V == 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w1 |= (uint32_t)bi[4 + e] << (8 * e);
        *reinterpret_cast<uint2*>(idx + o) = make_uint2(w0, w1);
      } else {
        *reinterpret_cast<uint32_t*>(idx + o) = w0;
      }
    };
    load_col(col[0], wb - 1);
    load_col(col[1], wb);
    int w = wb;
    for (; w + 3 <= we; w += 3) {
      load_col(col[2], w + 1); emit(col[0], col[1], col[2], w);
      load_col(col[0], w + 2); emit(col[1], col[2], col[0], w + 1);
      load_col(col[1], w + 3); emit(col[2], col[0], col[1], w + 2);
    }
    if (w < we) { load_col(col[2], w + 1); emit(col[0], col[1], col[2], w); ++w; }
    if (w < we) { load_col(col[0], w + 1); emit(col[1], col[2], col[0], w); }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s1_bwd_rows_kernel(const T* dy, const uint8_t* idx, T* dx, int N, int H, int W, int C, int SEG) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const int nseg = (W + SEG - 1) / SEG;
  const size_t rows = (size_t)N * H * nseg * CV;
  GRID_STRIDE(i, rows) {
    const int cv = (int)(i % CV);
    size_t t = i / CV;
    const int sg = (int)(t % nseg); t /= nseg;
    const int h = (int)(t % H), n = (int)(t / H);
    const int wb = sg * SEG, we = min(W, wb + SEG);
    // column slot: the windows centred in one column ow, rows oh = h + 1, h, h - 1 (tap row a = 0, 1, 2 of the pixel in row h)
    float g[3][3][V];
    uint32_t code[3][3][V / 4];
    auto load_col = [&](float (&gc)[3][V], uint32_t (&cc)[3][V / 4], int ow) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int oh = h + 1 - a;
        if (ow >= 0 && ow < W && oh >= 0 && oh < H) {
          const size_t o = (((size_t)n * H + oh) * W + ow) * C + cv * V;
          Vec16<T>::load(dy + o, gc[a]);
          if constexpr (V == 8) {
            const uint2 t2 = *reinterpret_cast<const uint2*>(idx + o);
            cc[a][0] = t2.x;
            cc[a][1] = t2.y;
          } else {
            cc[a][0] = *reinterpret_cast<const uint32_t*>(idx + o);
          }
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) gc[a][e] = 0.f;
#pragma unroll
          for (int q = 0; q < V / 4; ++q) cc[a][q] = 0xffffffffu;  // (no tap has code 255)
        }
      }
    };
    // pixel w: column ow = w + 1 is tap column b = 0, ow = w is b = 1, ow = w - 1 is b = 2; additions in (a, b) order as the point-wise kernel
    auto emit = [&](const float (&gl)[3][V], const uint32_t (&cl)[3][V / 4], const float (&gm)[3][V], const uint32_t (&cm)[3][V / 4], const float (&gr)[3][V],
                    const uint32_t (&cr)[3][V / 4], int w) __attribute__((always_inline)) {
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const int sh = 8 * (e & 3);
          // (a skipped window of the point-wise form adds nothing; here it adds an exact 0.f: the same sum)
          acc[e] += ((cr[a][e >> 2] >> sh) & 0xffu) == (unsigned)(a * 3) ? gr[a][e] : 0.f;
          acc[e] += ((cm[a][e >> 2] >> sh) & 0xffu) == (unsigned)(a * 3 + 1) ? gm[a][e] : 0.f;
          acc[e] += ((cl[a][e >> 2] >> sh) & 0xffu) == (unsigned)(a * 3 + 2) ? gl[a][e] : 0.f;
        }
      }
      Vec16<T>::store(dx + ((((size_t)n * H + h) * W + w) * CV + cv) * V, acc);
    };
    load_col(g[0], code[0], wb - 1);
    load_col(g[1], code[1], wb);
    int w = wb;
    for (; w + 3 <= we; w += 3) {
      load_col(g[2], code[2], w + 1); emit(g[0], code[0], g[1], code[1], g[2], code[2], w);
      load_col(g[0], code[0], w + 2); emit(g[1], code[1], g[2], code[2], g[0], code[0], w + 1);
      load_col(g[1], code[1], w + 3); emit(g[2], code[2], g[0], code[0], g[1], code[1], w + 2);
    }
    if (w < we) { load_col(g[2], code[2], w + 1); emit(g[0], code[0], g[1], code[1], g[2], code[2], w); ++w; }
    if (w < we) { load_col(g[0], code[0], w + 1); emit(g[1], code[1], g[2], code[2], g[0], code[0], w); }
  }
}

// ---- ECA ------------------------------------------------------------------------------------------------------------
// gate[n][c] = sigmoid(sum_j w[j] * pooled[n][c + j - k/2])
__global__ void eca_gate_kernel(const float* pooled, const float* w, int k, float* gate, int N, int C) {
  GRID_STRIDE(i, (size_t)N * C) {
    const int c = (int)(i % C), n = (int)(i / C);
    float z = 0.f;
    for (int j = 0; j < k; ++j) {
      const int cc = c + j - k / 2;
      if (cc >= 0 && cc < C) z += w[j] * pooled[(size_t)n * C + cc];
    }
    gate[i] = 1.f / (1.f + __expf(-z));
  }
}
// the same with the BatchNorm the pooled means still have to go through (pooled = raw * xs[c] + xh[c]: pooled_affine_kernel) applied on the way: one
// launch instead of two, raw and pooled are different arrays (a neighbour's raw value is read after its owner has written the affine one otherwise)
__global__ void eca_gate_affine_kernel(const float* raw, const float* xs, const float* xh, const float* w, int k, float* pooled, float* gate, int N, int C) {
  GRID_STRIDE(i, (size_t)N * C) {
    const int c = (int)(i % C), n = (int)(i / C);
    float z = 0.f;
    for (int j = 0; j < k; ++j) {
      const int cc = c + j - k / 2;
      if (cc >= 0 && cc < C) z += w[j] * fmaf(raw[(size_t)n * C + cc], xs[cc], xh[cc]);
    }
    pooled[i] = fmaf(raw[i], xs[c], xh[c]);
    gate[i] = 1.f / (1.f + __expf(-z));
  }
}
// y = x * gate[n][c]   (BWD: dx = dy * gate[n][c] + dpool[n][c])
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void eca_scale_kernel(const T* x, const float* gate, const float* dpool, T* y, int N, int HW, int C) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const size_t total = (size_t)N * HW * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    const int n = (int)(i / ((size_t)HW * CV));
    float v[V];
    Vec16<T>::load(x + i * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const size_t g = (size_t)n * C + cv * V + e;
      v[e] = v[e] * gate[g] + (BWD ? dpool[g] : 0.f);
    }
    Vec16<T>::store(y + i * V, v);
  }
}
// s[n][c] = sum_hw dy * x : one workgroup per (image, slab of 8 channel vectors), 256 threads = 32 pixel lanes x 8 vectors of
// 16 bytes; fixed-order accumulation + fixed-shape LDS tree (as gap_fwd_kernel)
template <typename T>
__global__ __launch_bounds__(256) void eca_prod_reduce_kernel(const T* dy, const T* x, float* s, int N, int HW, int C) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[32][8 * V + 1];
  const int slabs = C / (8 * V);
  const int n = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * 8 * V;
  const int cv = threadIdx.x & 7, r = threadIdx.x >> 3;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  for (int p = r; p < HW; p += 32) {
    const size_t o = ((size_t)n * HW + p) * C + c0 + cv * V;
    float a[V], b[V];
    Vec16<T>::load(dy + o, a);
    Vec16<T>::load(x + o, b);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] += a[e] * b[e];
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[r][cv * V + e] = acc[e];
  __syncthreads();
  for (int st = 16; st > 0; st >>= 1) {
    if (r < st) {
#pragma unroll
      for (int e = 0; e < V; ++e) red[r][cv * V + e] += red[r + st][cv * V + e];
    }
    __syncthreads();
  }
  if (r == 0) {
#pragma unroll
    for (int e = 0; e < V; ++e) s[(size_t)n * C + c0 + cv * V + e] = red[0][cv * V + e];
  }
}
// dpre = s * g * (1 - g); dpool[n][c] = (sum_j w[j] * dpre[n][c - j + k/2]) / HW; dw[j] = sum_{n,c} dpre[n][c] * pooled[n][c + j - k/2]
// The tensors are [N][C] (up to 256 x 2048): ECA_GB workgroups walk them grid-stride, each leaves its k partial weight
// gradients in dwpart[block][9]; eca_dw_finish adds the ECA_GB rows in block order (fixed grid => bitwise reproducible).
constexpr int ECA_GB = 128;
// ECA_BT threads per workgroup: the kernel is a chain of dependent loads per element (29 us per launch with 256 threads = 16 elements per thread at
// 256 x 2048; 16 launches per BResNet-50 step on the caller's stream) — 1024 threads walk 4 elements each; still ECA_GB partial rows (the scratch contract).
constexpr int ECA_BT = 1024;
__global__ __launch_bounds__(ECA_BT) void eca_gate_bwd_kernel(const float* s, const float* gate, const float* pooled, const float* w, int k,
                                                             float* dpool, float* dwpart, int N, int C, float inv_hw) {
  __shared__ float red[ECA_BT];
  float part[9];
  for (int j = 0; j < 9; ++j) part[j] = 0.f;
  const size_t total = (size_t)N * C;
  for (size_t i = (size_t)blockIdx.x * ECA_BT + threadIdx.x; i < total; i += (size_t)ECA_GB * ECA_BT) {
    const int c = (int)(i % C), n = (int)(i / C);
    const float dpre_i = s[i] * gate[i] * (1.f - gate[i]);
    float dp = 0.f;
    for (int j = 0; j < k; ++j) {
      const int cc = c - j + k / 2;  // dpre element that read pooled[n][c] through tap j
      if (cc >= 0 && cc < C) {
        const size_t o = (size_t)n * C + cc;
        dp += w[j] * s[o] * gate[o] * (1.f - gate[o]);
      }
      const int cp = c + j - k / 2;
      if (cp >= 0 && cp < C) part[j] += dpre_i * pooled[(size_t)n * C + cp];
    }
    dpool[i] = dp * inv_hw;
  }
  for (int j = 0; j < k; ++j) {
    red[threadIdx.x] = part[j];
    __syncthreads();
    for (int st = ECA_BT / 2; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) dwpart[blockIdx.x * 9 + j] = red[0];
    __syncthreads();
  }
}
__global__ void eca_dw_finish_kernel(const float* dwpart, float* dw, int k, float beta) {
  const int j = threadIdx.x;
  if (j >= k) return;
  float a = 0.f;
  for (int b = 0; b < ECA_GB; ++b) a += dwpart[b * 9 + j];
  dw[j] = (beta != 0.f ? beta * dw[j] : 0.f) + a;
}

// ---- ECA + drop-connect + shortcut + activation in one pass each way (the static executor, bresnet_exec.cpp) ------------------------
// Forward: out = act(x * gate[n][c] * keep[n] + shortcut) — the gated tensor (eca_scale) is never stored.
// xs / xh (optional): x stands for x * xs[c] + xh[c], i.e. the raw conv output under its BatchNorm (identity activation) applied on the fly —
// the normalised tensor is never stored either; ss / sh2 the same for the shortcut (the downsample branch's BatchNorm).
struct Affine {
  const float* xs;
  const float* xh;
  const float* ss;
  const float* sh2;
};
template <typename T>
__global__ __launch_bounds__(256) void eca_residual_fwd_kernel(const T* x, const float* gate, const float* keep, const T* shortcut, T* out, int N, int HW,
                                                               int C, int act, Affine af, uint8_t* bits) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const size_t total = (size_t)N * HW * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    const int n = (int)(i / ((size_t)HW * CV));
    const float kn = keep ? keep[n] : 1.f;
    float v[V], s[V];
    Vec16<T>::load(x + i * V, v);
    Vec16<T>::load(shortcut + i * V, s);
    if (af.xs) {
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = fmaf(v[e], af.xs[cv * V + e], af.xh[cv * V + e]);
    }
    if (af.ss) {
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] = fmaf(s[e], af.ss[cv * V + e], af.sh2[cv * V + e]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = act_fwd(v[e] * gate[(size_t)n * C + cv * V + e] * kn + s[e], act);
    Vec16<T>::store(out + i * V, v);
    if (bits) {  // the sign of the STORED value, one byte per 16-byte vector: what the backward's activation slope needs of `out`
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < V; ++e) b |= ((float)(T)v[e] > 0.f ? 1u : 0u) << e;
      __builtin_nontemporal_store((uint8_t)b, bits + i);
    }
  }
}
// Backward, pass 1: dz = dout * act'(out) is the shortcut gradient (stored) and, times keep[n], the gradient of the gated tensor, whose
// product with x summed over the pixels is what the gate's backward needs: s[n][c] = sum_hw dz * keep[n] * x.  Work layout and summation
// order of eca_prod_reduce_kernel.
// SUMS: also a[n][c] = sum_hw dz, b[n][c] = sum_hw dz * x (x raw: the conv output under the on-the-fly BatchNorm), y[n][c] = sum_hw x into
// sums[0 .. 2][N][C] — everything the BatchNorm backward of x's layer needs from the big tensors (eca_bn_sums_kernel below)
// x2 (SUMS only, optional): the raw output of the downsample convolution, whose BatchNorm (identity activation) has dz itself as the gradient of its
// output: d[n][c] = sum_hw dz * x2 into sums[3] — with a[n][c] all that layer's backward sums need (eca_ds_sums_kernel)
template <typename T, bool SUMS>
__global__ __launch_bounds__(256) void eca_residual_bwd_reduce_kernel(const T* dout, const T* out, const T* x, const float* keep, T* dshortcut, float* s, int N,
                                                                      int HW, int C, int act, const float* xs, const float* xh, float* sums, const uint8_t* bits,
                                                                      const T* x2) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[32][8 * V + 1];
  const int slabs = C / (8 * V);
  const int n = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * 8 * V;
  const int cv = threadIdx.x & 7, r = threadIdx.x >> 3;
  const float kn = keep ? keep[n] : 1.f;
  float acc[V], sc[V], sh[V], sa[V], sb[V], sy[V], sd[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    acc[e] = sa[e] = sb[e] = sy[e] = sd[e] = 0.f;
    sc[e] = xs ? xs[c0 + cv * V + e] : 1.f;
    sh[e] = xs ? xh[c0 + cv * V + e] : 0.f;
  }
  for (int p = r; p < HW; p += 32) {
    const size_t o = ((size_t)n * HW + p) * C + c0 + cv * V;
    float g[V], ov[V], xv[V];
    Vec16<T>::load(dout + o, g);
    if (bits) {  // (uniform branch) the forward's sign bits stand for `out`: 1 byte instead of 16
      const unsigned b = bits[o / V];
#pragma unroll
      for (int e = 0; e < V; ++e) ov[e] = (b >> e) & 1u ? 1.f : -1.f;
    } else {
      Vec16<T>::load(out + o, ov);
    }
    Vec16<T>::load(x + o, xv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      g[e] *= act_slope(ov[e], act);
      acc[e] += g[e] * kn * fmaf(xv[e], sc[e], sh[e]);
      if constexpr (SUMS) {
        const float gr = (float)(T)g[e];  // the value stored below: what the later passes read
        sa[e] += gr;
        sb[e] = fmaf(gr, xv[e], sb[e]);
        sy[e] += xv[e];
      }
    }
    if constexpr (SUMS) {
      if (x2) {  // (uniform branch)
        float dv[V];
        Vec16<T>::load(x2 + o, dv);
#pragma unroll
        for (int e = 0; e < V; ++e) sd[e] = fmaf((float)(T)g[e], dv[e], sd[e]);
      }
    }
    Vec16<T>::store(dshortcut + o, g);
  }
  // rows r = 8w .. 8w + 7 of a channel vector sit in one wave (lane = (r & 7) * 8 + cv): shuffles across them, then the four waves through LDS
  auto wave_sum = [&](float (&v)[V]) __attribute__((always_inline)) {
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] += __shfl_xor(v[e], off);
    }
  };
  wave_sum(acc);
  if constexpr (SUMS) {
    wave_sum(sa);
    wave_sum(sb);
    wave_sum(sy);
    wave_sum(sd);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 8) {  // (rows 4 * k of `red`: array k; columns: wave, channel)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red[wv][cv * V + e] = acc[e];
      if constexpr (SUMS) {
        red[4 + wv][cv * V + e] = sa[e];
        red[8 + wv][cv * V + e] = sb[e];
        red[12 + wv][cv * V + e] = sy[e];
        red[16 + wv][cv * V + e] = sd[e];
      }
    }
  }
  __syncthreads();
  const int NARR = SUMS ? (x2 ? 5 : 4) : 1;
  for (int i = threadIdx.x; i < NARR * 8 * V; i += 256) {
    const int k = i / (8 * V), ch = i % (8 * V);
    const float t = (red[4 * k][ch] + red[4 * k + 1][ch]) + (red[4 * k + 2][ch] + red[4 * k + 3][ch]);
    float* dst = k == 0 ? s : sums + (size_t)(k - 1) * N * C;
    dst[(size_t)n * C + c0 + ch] = t;
  }
}
// BatchNorm-backward sums of the layer under the ECA module WITHOUT a second pass over the tensors: with dz3 = dz * keep[n] * gate[n][c] + dpool[n][c],
//   sum dz3        = sum_n keep gate a + HW dpool
//   sum dz3 * xhat = invstd * sum_n [ keep gate (b - mean a) + dpool (y - HW mean) ]          (xhat = (x - mean) * invstd)
// from the per-image sums of pass 1 — one partial row [2][C] for bn_finalize (fp64 over the images, in image order)
constexpr int ECA_SL = 64;   // image slices per channel (1024 threads per workgroup: 4 images per thread at batch 256 — one round of loads instead of four dependent ones)
__global__ __launch_bounds__(16 * ECA_SL) void eca_bn_sums_kernel(const float* sums, const float* gate, const float* dpool, const float* keep, const float* mean,
                                                                 const float* invstd, float* row, int N, int C, int HW) {
  // 16 channels x ECA_SL image slices per workgroup (a serial walk over the images is a chain of 256 load latencies); slices added in slice order
  __shared__ double red[2][ECA_SL][16];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const double mu = (double)mean[c], hw = (double)HW;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int n = sl; n < N; n += ECA_SL) {
    const size_t o = (size_t)n * C + c;
    const double kg = (double)(keep ? keep[n] : 1.f) * (double)gate[o], dp = (double)dpool[o];
    const double a = (double)sums[o], b = (double)sums[(size_t)N * C + o], y = (double)sums[(size_t)2 * N * C + o];
    s1 += kg * a + hw * dp;
    s2 += kg * (b - mu * a) + dp * (y - hw * mu);
  }
  red[0][sl][cl] = s1;
  red[1][sl][cl] = s2;
  __syncthreads();
  if (sl == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < ECA_SL; ++k) {
      t1 += red[0][k][cl];
      t2 += red[1][k][cl];
    }
    row[c] = (float)t1;
    row[C + c] = (float)(t2 * (double)invstd[c]);
  }
}
// the downsample BatchNorm's backward sums (identity activation: its output gradient is dz): sum dz = sum_n a, sum dz * xhat = invstd (sum_n d - mean sum_n a)
__global__ __launch_bounds__(16 * ECA_SL) void eca_ds_sums_kernel(const float* sums, const float* mean, const float* invstd, float* row, int N, int C) {
  __shared__ double red[2][ECA_SL][16];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int n = sl; n < N; n += ECA_SL) {
    const size_t o = (size_t)n * C + c;
    s1 += (double)sums[o];
    s2 += (double)sums[(size_t)3 * N * C + o];
  }
  red[0][sl][cl] = s1;
  red[1][sl][cl] = s2;
  __syncthreads();
  if (sl == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < ECA_SL; ++k) {
      t1 += red[0][k][cl];
      t2 += red[1][k][cl];
    }
    row[c] = (float)t1;
    row[C + c] = (float)((t2 - (double)mean[c] * t1) * (double)invstd[c]);
  }
}
// pass 2: dx = dz * keep[n] * gate[n][c] + dpool[n][c]
template <typename T>
__global__ __launch_bounds__(256) void eca_residual_bwd_apply_kernel(const T* dz, const float* gate, const float* dpool, const float* keep, T* dx, int N, int HW,
                                                                     int C) {
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const size_t total = (size_t)N * HW * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    const int n = (int)(i / ((size_t)HW * CV));
    const float kn = keep ? keep[n] : 1.f;
    float v[V];
    Vec16<T>::load(dz + i * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const size_t g = (size_t)n * C + cv * V + e;
      v[e] = v[e] * kn * gate[g] + dpool[g];
    }
    Vec16<T>::store(dx + i * V, v);
  }
}

// pooled[n][c] = pooled[n][c] * xs[c] + xh[c]: the pooled features of a tensor that only exists as (raw conv output, BN scale / shift)
__global__ void pooled_affine_kernel(float* pooled, const float* xs, const float* xh, int N, int C) {
  GRID_STRIDE(i, (size_t)N * C) {
    const int c = (int)(i % C);
    pooled[i] = fmaf(pooled[i], xs[c], xh[c]);
  }
}

// ---- weight standardisation: one workgroup per output channel ---------------------------------------------------------
__global__ __launch_bounds__(256) void weight_std_fwd_kernel(const float* w, float* w_hat, float* mean, float* invstd, int K, float eps) {
  __shared__ double r1[256], r2[256];
  const float* row = w + (size_t)blockIdx.x * K;
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < K; i += 256) { a += row[i]; b += (double)row[i] * row[i]; }
  r1[threadIdx.x] = a; r2[threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
    __syncthreads();
  }
  const double mu = r1[0] / K, var = fmax(r2[0] / K - mu * mu, 0.0);
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) { mean[blockIdx.x] = (float)mu; invstd[blockIdx.x] = is; }
  for (int i = threadIdx.x; i < K; i += 256) w_hat[(size_t)blockIdx.x * K + i] = (row[i] - (float)mu) * is;
}
// dw = invstd * (g - mean(g) - w_hat * mean(g * w_hat))
__global__ __launch_bounds__(256) void weight_std_bwd_kernel(const float* g, const float* w_hat, const float* invstd, float* dw, float beta, int K) {
  __shared__ double r1[256], r2[256];
  const size_t base = (size_t)blockIdx.x * K;
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < K; i += 256) { a += g[base + i]; b += (double)g[base + i] * w_hat[base + i]; }
  r1[threadIdx.x] = a; r2[threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
    __syncthreads();
  }
  const float m1 = (float)(r1[0] / K), m2 = (float)(r2[0] / K), is = invstd[blockIdx.x];
  for (int i = threadIdx.x; i < K; i += 256) {
    const float v = is * (g[base + i] - m1 - w_hat[base + i] * m2);
    dw[base + i] = (beta != 0.f ? beta * dw[base + i] : 0.f) + v;
  }
}

// ---- residual + activation -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void residual_act_fwd_kernel(const T* branch, const float* scale_n, const T* shortcut, T* out, int N, size_t HWC, int act) {
  constexpr int V = Vec16<T>::N;
  const size_t per = HWC / V, total = (size_t)N * per;
  GRID_STRIDE(i, total) {
    const float sc = scale_n ? scale_n[i / per] : 1.f;
    float b[V], s[V];
    Vec16<T>::load(branch + i * V, b);
    if (shortcut) Vec16<T>::load(shortcut + i * V, s);
#pragma unroll
    for (int e = 0; e < V; ++e) b[e] = act_fwd(b[e] * sc + (shortcut ? s[e] : 0.f), act);
    Vec16<T>::store(out + i * V, b);
  }
}
// dz = dout * act'(out);  dbranch = dz * scale[n];  dshortcut = dz
template <typename T>
__global__ __launch_bounds__(256) void residual_act_bwd_kernel(const T* dout, const T* out, const float* scale_n, T* dbranch, T* dshortcut, int N, size_t HWC,
                                                               int act) {
  constexpr int V = Vec16<T>::N;
  const size_t per = HWC / V, total = (size_t)N * per;
  GRID_STRIDE(i, total) {
    const float sc = scale_n ? scale_n[i / per] : 1.f;
    float g[V], o[V], db[V];
    Vec16<T>::load(dout + i * V, g);
    Vec16<T>::load(out + i * V, o);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      g[e] *= act_slope(o[e], act);
      db[e] = g[e] * sc;
    }
    Vec16<T>::store(dbranch + i * V, db);
    if (dshortcut) Vec16<T>::store(dshortcut + i * V, g);
  }
}

// ---- dropout / drop-connect masks from a counter-based generator -----------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// keep[i] = u_i >= p ? 1 / (1 - p) : 0
__global__ void keep_scale_kernel(float* keep, size_t n, float p, unsigned long long seed, unsigned long long counter) {
  const unsigned long long base = splitmix64(seed ^ splitmix64(counter));
  GRID_STRIDE(i, n) {
    const float u = ((float)(splitmix64(base + 0x632BE59BD9B4E019ull * (i + 1)) >> 40) + 0.5f) * (1.0f / 16777216.0f);
    keep[i] = u >= p ? 1.f / (1.f - p) : 0.f;
  }
}
// every drop-connect / dropout scale array of a forward pass in one launch (blockIdx.y = array): the same counter-based values as keep_scale_kernel
__global__ void keep_scale_batch_kernel(const KeepBatch kb, unsigned long long seed) {
  const int e = blockIdx.y;
  const unsigned long long base = splitmix64(seed ^ splitmix64(kb.counter[e]));
  const float p = kb.p[e];
  float* keep = kb.keep[e];
  GRID_STRIDE(i, kb.n[e]) {
    const float u = ((float)(splitmix64(base + 0x632BE59BD9B4E019ull * (i + 1)) >> 40) + 0.5f) * (1.0f / 16777216.0f);
    keep[i] = u >= p ? 1.f / (1.f - p) : 0.f;
  }
}
__global__ void mul_kernel(const float* a, const float* b, float* out, size_t n) {
  GRID_STRIDE(i, n) out[i] = a[i] * b[i];
}

// ---- glue of the static BResNet-50 executor (bresnet_exec.cpp): what the per-op graph did with torch cat / permute / .to() --------
// NCHW fp32 batch -> NHWC with 64 channels (3 real, 61 zero: the granule of the conv kernels): one thread per pixel and 16-byte vector
template <typename T>
__global__ __launch_bounds__(256) void nchw_pad64_kernel(const float* __restrict__ x, T* __restrict__ h, int N, int HW) {
  constexpr int V = Vec16<T>::N, CV = 64 / V;
  const size_t total = (size_t)N * HW * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    const size_t pix = i / CV;
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    if (cv == 0) {
      const size_t n = pix / HW, q = pix - n * HW;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = x[(n * 3 + c) * HW + q];
    }
    Vec16<T>::store(h + pix * 64 + cv * V, v);
  }
}

// NCHW fp32 batch -> the patches of a 3x3 / stride-2 / pad-1 convolution, NHWC on the OUTPUT grid: h[n][oh][ow][(kh*3 + kw)*3 + c] = x[n][c][2oh + kh - 1][2ow + kw - 1]
// (zero outside the image), 27 real of 64 channels.  The deep stem's first convolution (3 -> 32, 3x3 / 2) is then a 1x1 convolution over this tensor with
// its own weights [32][3][3][3] read as [32][27]: 411 MB written and read at batch 256 instead of the 1.64 GB of the 64-channel full-resolution input.
// One thread per output pixel and 16-byte vector.
template <typename T>
__global__ __launch_bounds__(256) void nchw_im2col3s2_kernel(const float* __restrict__ x, T* __restrict__ h, int N, int H, int W) {
  constexpr int V = Vec16<T>::N, CV = 64 / V;
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo * CV;
  GRID_STRIDE(i, total) {
    const int cv = (int)(i % CV);
    const size_t pix = i / CV;
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    if (cv * V < 27) {
      const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho);
      const size_t n = pix / ((size_t)Wo * Ho);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const int k = cv * V + e;
        if (k < 27) {
          const int c = k % 3, t = k / 3, kh = t / 3, kw = t % 3;
          const int ih = 2 * oh + kh - 1, iw = 2 * ow + kw - 1;
          if (ih >= 0 && ih < H && iw >= 0 && iw < W) v[e] = x[((n * 3 + c) * H + ih) * (size_t)W + iw];
        }
      }
    }
    Vec16<T>::store(h + pix * 64 + cv * V, v);
  }
}

// conv weights fp32 [Cout][taps][Cin] -> T [Coutp][taps][Cinp], zero padded
template <typename T>
__global__ __launch_bounds__(256) void weight_pad_cast_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cout, int taps, int Cin, int Coutp, int Cinp) {
  const size_t total = (size_t)Coutp * taps * Cinp;
  GRID_STRIDE(i, total) {
    const int ci = (int)(i % Cinp);
    const size_t r = i / Cinp;
    const int t = (int)(r % taps), co = (int)(r / taps);
    wp[i] = (T)((co < Cout && ci < Cin) ? w[((size_t)co * taps + t) * Cin + ci] : 0.f);
  }
}

// every convolution's [standardise ->] pad -> cast in ONE launch: workgroup b = padded output row (layer l, channel co) of the table; the reduction is
// weight_std_fwd_kernel's (same strided partial sums, same tree), the values written are pad_cast's of the standardised row
template <typename T>
__global__ __launch_bounds__(256) void bres_weight_rows_kernel(const BPrepDesc* __restrict__ table, int nconv, float eps) {
  __shared__ double r1[256], r2[256];
  const int b = blockIdx.x;
  int l = 0;
  while (l + 1 < nconv && table[l + 1].row_begin <= b) ++l;   // uniform
  const BPrepDesc d = table[l];
  const int co = b - d.row_begin;
  const int K = d.taps * d.Cin;
  T* wp = reinterpret_cast<T*>(d.wp) + (size_t)co * d.taps * d.Cinp;
  if (co >= d.Cout) {   // a padding row
    for (int i = threadIdx.x; i < d.taps * d.Cinp; i += 256) wp[i] = (T)0.f;
    return;
  }
  const float* row = d.w + (size_t)co * K;
  float mu = 0.f, is = 1.f;
  if (d.w_hat) {
    double a = 0, q = 0;
    for (int i = threadIdx.x; i < K; i += 256) { a += row[i]; q += (double)row[i] * row[i]; }
    r1[threadIdx.x] = a; r2[threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
      __syncthreads();
    }
    const double m = r1[0] / K, var = fmax(r2[0] / K - m * m, 0.0);
    mu = (float)m;
    is = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { d.mean[co] = mu; d.invstd[co] = is; }
    for (int i = threadIdx.x; i < K; i += 256) d.w_hat[(size_t)co * K + i] = (row[i] - mu) * is;
  }
  for (int i = threadIdx.x; i < d.taps * d.Cinp; i += 256) {
    const int t = i / d.Cinp, ci = i - t * d.Cinp;
    float v = 0.f;
    if (ci < d.Cin) {
      v = row[t * d.Cin + ci];
      if (d.w_hat) v = (v - mu) * is;
    }
    wp[i] = (T)v;
  }
}

// ... and every transposed copy wtr[ci][t][co] = wp[co][t][ci] in one launch of 32 x 32 tiles (weights.hip weight_prep_kernel's tile)
template <typename T>
__global__ __launch_bounds__(256) void bres_weight_transpose_kernel(const BPrepDesc* __restrict__ table, int nconv) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int l = 0;
  while (l + 1 < nconv && table[l + 1].tile_begin <= b) ++l;
  const BPrepDesc d = table[l];
  const int local = b - d.tile_begin;
  const int nci = d.Cinp / 32, nco = d.Coutp / 32;
  const int ci0 = (local % nci) * 32, co0 = ((local / nci) % nco) * 32, t = local / (nci * nco);
  const T* w = reinterpret_cast<const T*>(d.wp);
  T* wt = reinterpret_cast<T*>(d.wtr);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) tile[r][tx] = (float)w[((size_t)(co0 + r) * d.taps + t) * d.Cinp + ci0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) wt[((size_t)(ci0 + r) * d.taps + t) * d.Coutp + co0 + tx] = (T)tile[tx][r];
}

// ... and the way back for the gradient: dw[Cout][taps][Cin] = beta * dw + dwp[Coutp][taps][Cinp] restricted
__global__ __launch_bounds__(256) void weight_unpad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, float beta, int Cout, int taps, int Cin, int Cinp) {
  const size_t total = (size_t)Cout * taps * Cin;
  GRID_STRIDE(i, total) {
    const int ci = (int)(i % Cin);
    const size_t r = i / Cin;  // co * taps + t
    const float g = dwp[r * Cinp + ci];
    dw[i] = beta != 0.f ? beta * dw[i] + g : g;
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ src, float* __restrict__ dst, float beta, size_t n) {
  GRID_STRIDE(i, n) dst[i] = beta != 0.f ? beta * dst[i] + src[i] : src[i];
}

}  // namespace
}  // namespace mi355

namespace mi355 {
int launch_nchw_pad64(int dtype, const float* x, void* h, int N, int HW, hipStream_t s) {
  const size_t total = (size_t)N * HW * 64;
  if (dtype == MI355_F32) hipLaunchKernelGGL(nchw_pad64_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, x, (float*)h, N, HW);
  else hipLaunchKernelGGL(nchw_pad64_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, x, (bf16_t*)h, N, HW);
  MI355_LAUNCH_CHECK();
  return 0;
}
int launch_nchw_im2col3s2(int dtype, const float* x, void* h, int N, int H, int W, hipStream_t s) {
  MI355_ARG(H % 2 == 0 && W % 2 == 0, "nchw_im2col3s2: H=%d W=%d (even)", H, W);
  const size_t total = (size_t)N * (H / 2) * (W / 2) * 64;
  if (dtype == MI355_F32) hipLaunchKernelGGL(nchw_im2col3s2_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, x, (float*)h, N, H, W);
  else hipLaunchKernelGGL(nchw_im2col3s2_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, x, (bf16_t*)h, N, H, W);
  MI355_LAUNCH_CHECK();
  return 0;
}
int launch_weight_pad_cast(int dtype, const float* w, void* wp, int Cout, int taps, int Cin, int Coutp, int Cinp, hipStream_t s) {
  const size_t total = (size_t)Coutp * taps * Cinp;
  if (dtype == MI355_F32) hipLaunchKernelGGL(weight_pad_cast_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)wp, Cout, taps, Cin, Coutp, Cinp);
  else hipLaunchKernelGGL(weight_pad_cast_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16_t*)wp, Cout, taps, Cin, Coutp, Cinp);
  MI355_LAUNCH_CHECK();
  return 0;
}
int launch_keep_scale_batch(const KeepBatch& kb, unsigned long long seed, hipStream_t s) {
  if (kb.count <= 0) return 0;
  size_t nmax = 0;
  for (int e = 0; e < kb.count; ++e) nmax = kb.n[e] > nmax ? kb.n[e] : nmax;
  hipLaunchKernelGGL(keep_scale_batch_kernel, dim3(grid_for(nmax), kb.count), dim3(256), 0, s, kb, seed);
  MI355_LAUNCH_CHECK();
  return 0;
}
int launch_bres_weight_prep(int dtype, const BPrepDesc* table, int nconv, int total_rows, int total_tiles, float eps, bool transposed, hipStream_t s) {
  MI355_ARG(table && nconv > 0 && total_rows > 0, "bres_weight_prep: empty table");
  if (dtype == MI355_F32) hipLaunchKernelGGL(bres_weight_rows_kernel<float>, dim3(total_rows), dim3(256), 0, s, table, nconv, eps);
  else hipLaunchKernelGGL(bres_weight_rows_kernel<bf16_t>, dim3(total_rows), dim3(256), 0, s, table, nconv, eps);
  MI355_LAUNCH_CHECK();
  if (transposed) {
    if (dtype == MI355_F32) hipLaunchKernelGGL(bres_weight_transpose_kernel<float>, dim3(total_tiles), dim3(256), 0, s, table, nconv);
    else hipLaunchKernelGGL(bres_weight_transpose_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, s, table, nconv);
    MI355_LAUNCH_CHECK();
  }
  return 0;
}
int launch_weight_unpad(const float* dwp, float* dw, float beta, int Cout, int taps, int Cin, int Cinp, hipStream_t s) {
  hipLaunchKernelGGL(weight_unpad_kernel, dim3(grid_for((size_t)Cout * taps * Cin)), dim3(256), 0, s, dwp, dw, beta, Cout, taps, Cin, Cinp);
  MI355_LAUNCH_CHECK();
  return 0;
}
// out = act(eca(x) * keep[n] + shortcut); pooled / gate [N][C] are kept for backward (k = 3 ... 9 odd)
int launch_eca_residual_fwd(int dtype, const void* x, const float* w, int k, const float* keep, const void* shortcut, void* out, float* pooled, float* gate,
                            int N, int HW, int C, int act, hipStream_t s, const float* xs, const float* xh, const float* ss, const float* sh2, uint8_t* out_bits,
                            float* raw_ws) {
  if (xs && raw_ws) {   // raw means into the scratch, affine + gate in one launch
    MI355_TRY(mi355_gap_fwd(dtype, x, raw_ws, N, HW, C, s));
    hipLaunchKernelGGL(eca_gate_affine_kernel, dim3(grid_for((size_t)N * C)), dim3(256), 0, s, raw_ws, xs, xh, w, k, pooled, gate, N, C);
  } else {
    MI355_TRY(mi355_gap_fwd(dtype, x, pooled, N, HW, C, s));
    if (xs) hipLaunchKernelGGL(pooled_affine_kernel, dim3(grid_for((size_t)N * C)), dim3(256), 0, s, pooled, xs, xh, N, C);
    hipLaunchKernelGGL(eca_gate_kernel, dim3(grid_for((size_t)N * C)), dim3(256), 0, s, pooled, w, k, gate, N, C);
  }
  const Affine af{xs, xh, ss, sh2};
  const size_t total = (size_t)N * HW * C;
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(eca_residual_fwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)x, gate, keep, (const float*)shortcut, (float*)out, N, HW, C, act, af, out_bits);
  else
    hipLaunchKernelGGL(eca_residual_fwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)x, gate, keep, (const bf16_t*)shortcut, (bf16_t*)out, N, HW, C, act, af, out_bits);
  MI355_LAUNCH_CHECK();
  return 0;
}
// its backward from dout: dshortcut (the shortcut operand's gradient), dx (the ECA input's), dw[k] (beta 0 / 1); ws: 2*N*C + 1152 floats
int launch_eca_residual_bwd(int dtype, const void* dout, const void* out, const void* x, const float* keep, const float* w, int k, const float* pooled,
                            const float* gate, void* dshortcut, void* dx, float* dw, float beta, float* ws, int N, int HW, int C, int act, hipStream_t s,
                            const float* xs, const float* xh, float* bn_row, const float* bn_mean, const float* bn_invstd, const uint8_t* out_bits,
                            const void* ds_y, float* ds_row, const float* ds_mean, const float* ds_invstd) {
  // ds_y / ds_row (with bn_row; ws then holds 6*N*C + 1152 floats): the same for the downsample BatchNorm from the raw downsample conv output
  // out_bits (optional): the sign bits the forward left for `out` (then `out` itself is not read)
  // dx == nullptr: pass 2 is left to the consumer — dx = dshortcut * keep[n] * gate[n][c] + dpool[n][c] with dpool = ws + N * C (EcaGrad)
  // bn_row (with dx == nullptr and x given raw under xs / xh; ws then holds 5*N*C + 1152 floats): the BatchNorm-backward sums of x's layer as one
  // partial row [2][C], from the per-image sums of pass 1 (eca_bn_sums_kernel)
  float *sprod = ws, *dpool = ws + (size_t)N * C, *dwpart = ws + (size_t)2 * N * C, *sums = ws + (size_t)2 * N * C + 1152;
  MI355_ARG(!bn_row || (!dx && xs && xh && bn_mean && bn_invstd), "eca_residual_bwd: the BatchNorm sums need the raw tensor form and no stored dx");
  MI355_ARG(!ds_row || (bn_row && ds_y && ds_mean && ds_invstd), "eca_residual_bwd: the downsample sums come with the bn3 sums");
  if (dtype == MI355_F32) {
    if (bn_row)
      hipLaunchKernelGGL((eca_residual_bwd_reduce_kernel<float, true>), dim3(N * (C / 32)), dim3(256), 0, s, (const float*)dout, (const float*)out, (const float*)x, keep,
                         (float*)dshortcut, sprod, N, HW, C, act, xs, xh, sums, out_bits, (const float*)(ds_row ? ds_y : nullptr));
    else
      hipLaunchKernelGGL((eca_residual_bwd_reduce_kernel<float, false>), dim3(N * (C / 32)), dim3(256), 0, s, (const float*)dout, (const float*)out, (const float*)x, keep,
                         (float*)dshortcut, sprod, N, HW, C, act, xs, xh, nullptr, out_bits, nullptr);
  } else {
    if (bn_row)
      hipLaunchKernelGGL((eca_residual_bwd_reduce_kernel<bf16_t, true>), dim3(N * (C / 64)), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x,
                         keep, (bf16_t*)dshortcut, sprod, N, HW, C, act, xs, xh, sums, out_bits, (const bf16_t*)(ds_row ? ds_y : nullptr));
    else
      hipLaunchKernelGGL((eca_residual_bwd_reduce_kernel<bf16_t, false>), dim3(N * (C / 64)), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x,
                         keep, (bf16_t*)dshortcut, sprod, N, HW, C, act, xs, xh, nullptr, out_bits, nullptr);
  }
  hipLaunchKernelGGL(eca_gate_bwd_kernel, dim3(ECA_GB), dim3(ECA_BT), 0, s, sprod, gate, pooled, w, k, dpool, dwpart, N, C, 1.f / (float)HW);
  hipLaunchKernelGGL(eca_dw_finish_kernel, dim3(1), dim3(64), 0, s, dwpart, dw, k, beta);
  if (bn_row) hipLaunchKernelGGL(eca_bn_sums_kernel, dim3(C / 16), dim3(16 * ECA_SL), 0, s, sums, gate, dpool, keep, bn_mean, bn_invstd, bn_row, N, C, HW);
  if (ds_row) hipLaunchKernelGGL(eca_ds_sums_kernel, dim3(C / 16), dim3(16 * ECA_SL), 0, s, sums, ds_mean, ds_invstd, ds_row, N, C);
  if (!dx) {
    MI355_LAUNCH_CHECK();
    return 0;
  }
  const size_t total = (size_t)N * HW * C;
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(eca_residual_bwd_apply_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)dshortcut, gate, dpool, keep, (float*)dx, N, HW, C);
  else
    hipLaunchKernelGGL(eca_residual_bwd_apply_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dshortcut, gate, dpool, keep, (bf16_t*)dx, N, HW, C);
  MI355_LAUNCH_CHECK();
  return 0;
}
int launch_axpby(const float* src, float* dst, float beta, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, beta, n);
  MI355_LAUNCH_CHECK();
  return 0;
}
}  // namespace mi355

using namespace mi355;

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)                  \
  do {                                                          \
    if ((dtype) == MI355_F32) { CALL_F32; }                     \
    else if ((dtype) == MI355_BF16) { CALL_BF16; }              \
    else { set_error("bad dtype %d", (dtype)); return MI355_E_ARG; } \
  } while (0)

extern "C" {

int mi355_blurpool_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
  MI355_ARG(x && y && N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "blurpool: H=%d W=%d (even) C=%d (multiple of 8)", H, W, C);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(blurpool_fwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)x, (float*)y, N, H, W, C),
             hipLaunchKernelGGL(blurpool_fwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, N, H, W, C));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_blurpool_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  MI355_ARG(dy && dx && N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "blurpool: H=%d W=%d (even) C=%d (multiple of 8)", H, W, C);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(blurpool_bwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)dy, (float*)dx, N, H, W, C),
             hipLaunchKernelGGL(blurpool_bwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
  MI355_ARG(x && y && N > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "avgpool2: H=%d W=%d (even) C=%d (multiple of 8)", H, W, C);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((avgpool2_kernel<float, false>), dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)x, (float*)y, N, H, W, C),
             hipLaunchKernelGGL((avgpool2_kernel<bf16_t, false>), dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, N, H, W, C));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  MI355_ARG(dy && dx && N > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "avgpool2: H=%d W=%d (even) C=%d (multiple of 8)", H, W, C);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((avgpool2_kernel<float, true>), dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)dy, (float*)dx, N, H, W, C),
             hipLaunchKernelGGL((avgpool2_kernel<bf16_t, true>), dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_maxpool3s1_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  MI355_ARG(x && y && idx && N > 0 && H > 0 && W > 0 && C % 8 == 0, "maxpool3s1: C=%d (multiple of 8)", C);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W * C;
  if (W >= 16 && (size_t)N * H * C / 8 >= 16384 && pool_seg(W) > 0) {  // enough rows to fill the chip: the row-walking form (3 loads per pixel instead of 9)
    const int SEG = pool_seg(W);
    const size_t rows = (size_t)N * H * ((W + SEG - 1) / SEG) * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3s1_fwd_rows_kernel<float>, dim3(grid_for(rows / 4)), dim3(256), 0, s, (const float*)x, (float*)y, idx, N, H, W, C, SEG),
               hipLaunchKernelGGL(maxpool3s1_fwd_rows_kernel<bf16_t>, dim3(grid_for(rows / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, SEG));
    MI355_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3s1_fwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)x, (float*)y, idx, N, H, W, C),
             hipLaunchKernelGGL(maxpool3s1_fwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_maxpool3s1_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream) {
  MI355_ARG(dy && dx && idx && N > 0 && H > 0 && W > 0 && C % 8 == 0, "maxpool3s1: C=%d (multiple of 8)", C);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W * C;
  if (W >= 16 && (size_t)N * H * C / 8 >= 16384 && pool_seg(W) > 0) {
    const int SEG = pool_seg(W);
    const size_t rows = (size_t)N * H * ((W + SEG - 1) / SEG) * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3s1_bwd_rows_kernel<float>, dim3(grid_for(rows / 4)), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, N, H, W, C, SEG),
               hipLaunchKernelGGL(maxpool3s1_bwd_rows_kernel<bf16_t>, dim3(grid_for(rows / 8)), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, N, H, W, C, SEG));
    MI355_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3s1_bwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, N, H, W, C),
             hipLaunchKernelGGL(maxpool3s1_bwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, N, H, W, C));
  MI355_LAUNCH_CHECK();
  return 0;
}

int mi355_eca_fwd(int dtype, const void* x, const float* w, int k, void* y, float* pooled, float* gate, int N, int HW, int C, void* stream) {
  MI355_ARG(x && w && y && pooled && gate && k >= 1 && k <= 9 && (k & 1) && C % 64 == 0, "eca: k=%d (odd, <= 9) C=%d (multiple of 64)", k, C);
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(launch_gap_fwd(dtype, x, pooled, N, HW, C, s));
  hipLaunchKernelGGL(eca_gate_kernel, dim3(grid_for((size_t)N * C)), dim3(256), 0, s, pooled, w, k, gate, N, C);
  const size_t total = (size_t)N * HW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((eca_scale_kernel<float, false>), dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)x, gate, nullptr, (float*)y, N, HW, C),
             hipLaunchKernelGGL((eca_scale_kernel<bf16_t, false>), dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)x, gate, nullptr, (bf16_t*)y, N, HW, C));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_eca_bwd(int dtype, const void* dy, const void* x, const float* w, int k, const float* pooled, const float* gate, void* dx, float* dw,
                  float beta, float* ws, int N, int HW, int C, void* stream) {
  MI355_ARG(dy && x && w && pooled && gate && dx && dw && ws && k >= 1 && k <= 9 && (k & 1) && C % 64 == 0, "eca_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  float* sprod = ws;                    // [N][C]
  float* dpool = ws + (size_t)N * C;    // [N][C]
  float* dwpart = ws + (size_t)2 * N * C;  // [ECA_GB][9]
  DISPATCH_T(dtype, hipLaunchKernelGGL(eca_prod_reduce_kernel<float>, dim3(N * (C / 32)), dim3(256), 0, s, (const float*)dy, (const float*)x, sprod, N, HW, C),
             hipLaunchKernelGGL(eca_prod_reduce_kernel<bf16_t>, dim3(N * (C / 64)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, sprod, N, HW, C));
  hipLaunchKernelGGL(eca_gate_bwd_kernel, dim3(ECA_GB), dim3(ECA_BT), 0, s, sprod, gate, pooled, w, k, dpool, dwpart, N, C, 1.f / (float)HW);
  hipLaunchKernelGGL(eca_dw_finish_kernel, dim3(1), dim3(64), 0, s, dwpart, dw, k, beta);
  const size_t total = (size_t)N * HW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((eca_scale_kernel<float, true>), dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)dy, gate, dpool, (float*)dx, N, HW, C),
             hipLaunchKernelGGL((eca_scale_kernel<bf16_t, true>), dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dy, gate, dpool, (bf16_t*)dx, N, HW, C));
  MI355_LAUNCH_CHECK();
  return 0;
}

int mi355_weight_std_fwd(const float* w, float* w_hat, float* mean, float* invstd, int Cout, int K, float eps, void* stream) {
  MI355_ARG(w && w_hat && mean && invstd && Cout > 0 && K > 0, "weight_std: bad arguments");
  hipLaunchKernelGGL(weight_std_fwd_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, w_hat, mean, invstd, K, eps);
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_weight_std_bwd(const float* dw_hat, const float* w_hat, const float* invstd, float* dw, float beta, int Cout, int K, void* stream) {
  MI355_ARG(dw_hat && w_hat && invstd && dw && Cout > 0 && K > 0, "weight_std_bwd: bad arguments");
  hipLaunchKernelGGL(weight_std_bwd_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dw_hat, w_hat, invstd, dw, beta, K);
  MI355_LAUNCH_CHECK();
  return 0;
}

int mi355_residual_act_fwd(int dtype, const void* branch, const float* scale_n, const void* shortcut, void* out, int N, size_t HWC, int act, void* stream) {
  MI355_ARG(branch && out && N > 0 && HWC % 8 == 0 && act >= 0 && act <= 2, "residual_act: HWC=%zu (multiple of 8) act=%d", HWC, act);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * HWC;
  DISPATCH_T(dtype, hipLaunchKernelGGL(residual_act_fwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)branch, scale_n, (const float*)shortcut, (float*)out, N, HWC, act),
             hipLaunchKernelGGL(residual_act_fwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)branch, scale_n, (const bf16_t*)shortcut, (bf16_t*)out, N, HWC, act));
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_residual_act_bwd(int dtype, const void* dout, const void* out, const float* scale_n, void* dbranch, void* dshortcut, int N, size_t HWC, int act,
                           void* stream) {
  MI355_ARG(dout && out && dbranch && N > 0 && HWC % 8 == 0 && act >= 0 && act <= 2, "residual_act_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * HWC;
  DISPATCH_T(dtype, hipLaunchKernelGGL(residual_act_bwd_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, (const float*)dout, (const float*)out, scale_n, (float*)dbranch, (float*)dshortcut, N, HWC, act),
             hipLaunchKernelGGL(residual_act_bwd_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)out, scale_n, (bf16_t*)dbranch, (bf16_t*)dshortcut, N, HWC, act));
  MI355_LAUNCH_CHECK();
  return 0;
}

int mi355_keep_scale(float* keep, size_t n, float p, unsigned long long seed, unsigned long long counter, void* stream) {
  MI355_ARG(keep && n > 0 && p >= 0.f && p < 1.f, "keep_scale: p=%f must be in [0, 1)", p);
  hipLaunchKernelGGL(keep_scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, keep, n, p, seed, counter);
  MI355_LAUNCH_CHECK();
  return 0;
}
int mi355_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream) {
  MI355_ARG(a && b && out, "mul: null pointer");
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
