// misc.hip — the remaining HBM-bound pieces of the ResNet-50 step on gfx950:
//   stem ingest (NCHW fp32 -> zero-padded NHWC4), maxpool 3x3/2 fwd/bwd, global average pool fwd/bwd,
//   label-smoothed soft-target cross entropy fwd+bwd, fused SGD-momentum, FC bias/padding helpers.
// Reference call sites: sota_imagenet/dali_dataloader.py:113-123 (tensor contract of the batch),
// sota_imagenet/callbacks.py:316-317 (model / criterion / backward), arg_parser.py:136-142 (SGD, CE).
#include "common.h"
#include "vec.h"

namespace mi355 {
namespace {

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void stem_ingest_kernel(const float* __restrict__ x, T* __restrict__ xpad, int N, int H, int W, int Hp,
                                   int Wp) {
  const size_t total = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const size_t t = i / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const size_t plane = (size_t)H * W;
    const float* src = x + (size_t)n * 3 * plane + (size_t)h * W + w;
    const float r = src[0], g = src[plane], b = src[2 * plane];
    T* dst = xpad + (((size_t)n * Hp + h + STEM_PAD) * Wp + w + STEM_PAD) * 4;
    if constexpr (sizeof(T) == 4) {
      f32x4 v = {r, g, b, 0.f};
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      bf16x4 v = {(bf16_t)r, (bf16_t)g, (bf16_t)b, (bf16_t)0.f};
      *reinterpret_cast<bf16x4*>(dst) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N,
                                   int H, int W, int C, int Ho, int Wo) {
  constexpr int V = Vec16<T>::N;
  const int cv = C / V;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cv) * V;
    size_t t = i / cv;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best[V];
    int bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      best[e] = -INFINITY;
      bi[e] = -1;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        float v[V];
        Vec16<T>::load(x + (((size_t)n * H + ih) * W + iw) * C + c0, v);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = kh * 3 + kw;
          }
        }
      }
    }
    const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c0;
    Vec16<T>::store(y + o, best);
#pragma unroll
    for (int e = 0; e < V; ++e) idx[o + e] = (uint8_t)bi[e];
  }
}

// stem: p = maxpool3x3/2(relu(y*scale+shift)) straight from the raw conv output — the full-resolution activation is never
// written (backward only needs its ReLU mask: one byte per 16-byte vector, written here by the window that owns the
// pixel, (h>>1, w>>1), which always contains it).  Max/argmax semantics = maxpool_fwd_kernel on the rounded activation
// (rounding is monotonic, so max-then-round gives the same value).
template <typename T>
__global__ void bn_relu_maxpool_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                       const float* __restrict__ shift, T* __restrict__ p, uint8_t* __restrict__ idx,
                                       uint8_t* __restrict__ bits, int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int V = Vec16<T>::N;
  const int cv = C / V;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cvi = (int)(i % cv);
    const int c0 = cvi * V;
    size_t t = i / cv;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float sc[V], sh[V];
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c0 + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(shift + c0 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[4 * q + e] = a[e];
        sh[4 * q + e] = b[e];
      }
    }
    float best[V];
    int bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      best[e] = -INFINITY;
      bi[e] = -1;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const size_t pix = ((size_t)n * H + ih) * W + iw;
        float v[V];
        Vec16<T>::load(y + pix * C + c0, v);
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float a = fmaf(v[e], sc[e], sh[e]);
          m |= (a > 0.f ? 1u : 0u) << e;
          v[e] = (float)(T)(a > 0.f ? a : 0.f);  // the value the unfused path would have stored
          if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = kh * 3 + kw;
          }
        }
        if (kh >= 1 && kw >= 1) bits[pix * cv + cvi] = (uint8_t)m;  // pixels (2oh + kh-1, 2ow + kw-1): owned here
      }
    }
    const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c0;
    Vec16<T>::store(p + o, best);
    if constexpr (V == 4) {
      *reinterpret_cast<uint32_t*>(idx + o) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    } else {
      uint2 w;
      w.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
      w.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
      *reinterpret_cast<uint2*>(idx + o) = w;
    }
  }
}

// The same operator, one thread per 2 x 2 block of OUTPUT pixels (Ho, Wo even): the four windows share a 5 x 5 input patch, so
// 25 loads + BN/ReLU evaluations serve what four independent threads do with 36; a patch row is consumed as soon as it is loaded
// (40 live values), taps reach every window in the same (kh, kw) scan order as above => identical maxima and argmax codes.
template <typename T>
__global__ void bn_relu_maxpool2_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                        T* __restrict__ p, uint8_t* __restrict__ idx, uint8_t* __restrict__ bits, int N, int H, int W, int C,
                                        int Ho, int Wo) {
  constexpr int V = Vec16<T>::N;
  const int cv = C / V;
  const int Hb = Ho / 2, Wb = Wo / 2;
  const size_t total = (size_t)N * Hb * Wb * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cvi = (int)(i % cv);
    const int c0 = cvi * V;
    size_t t = i / cv;
    const int bw = (int)(t % Wb);
    t /= Wb;
    const int bh = (int)(t % Hb);
    const int n = (int)(t / Hb);
    float sc[V], sh[V];
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c0 + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(shift + c0 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[4 * q + e] = a[e];
        sh[4 * q + e] = b[e];
      }
    }
    float best[4][V];
    int bi[4][V];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < V; ++e) {
        best[w][e] = -INFINITY;
        bi[w][e] = -1;
      }
    const int ih0 = 4 * bh - 1, iw0 = 4 * bw - 1;  // top-left corner of the 5 x 5 patch
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int ih = ih0 + r;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int iw = iw0 + c;
        if ((unsigned)iw >= (unsigned)W) continue;
        const size_t pix = ((size_t)n * H + ih) * W + iw;
        float v[V];
        Vec16<T>::load(y + pix * C + c0, v);
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float a = fmaf(v[e], sc[e], sh[e]);
          m |= (a > 0.f ? 1u : 0u) << e;
          v[e] = (float)(T)(a > 0.f ? a : 0.f);  // the value the unfused path would have stored
        }
        if (r >= 1 && c >= 1) bits[pix * cv + cvi] = (uint8_t)m;  // rows / columns 1..4 of the patch are owned by this block
        // windows (dr, dc) of the block that contain patch element (r, c): kh = r - 2 dr, kw = c - 2 dc in 0..2
#pragma unroll
        for (int dr = 0; dr < 2; ++dr) {
          const int kh = r - 2 * dr;
          if (kh < 0 || kh > 2) continue;
#pragma unroll
          for (int dc = 0; dc < 2; ++dc) {
            const int kw = c - 2 * dc;
            if (kw < 0 || kw > 2) continue;
            const int w = 2 * dr + dc;
#pragma unroll
            for (int e = 0; e < V; ++e)
              if (bi[w][e] < 0 || v[e] > best[w][e] || v[e] != v[e]) {
                best[w][e] = v[e];
                bi[w][e] = kh * 3 + kw;
              }
          }
        }
      }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const size_t o = (((size_t)n * Ho + 2 * bh + (w >> 1)) * Wo + 2 * bw + (w & 1)) * C + c0;
      Vec16<T>::store(p + o, best[w]);
      if constexpr (V == 4) {
        *reinterpret_cast<uint32_t*>(idx + o) = (uint32_t)bi[w][0] | ((uint32_t)bi[w][1] << 8) | ((uint32_t)bi[w][2] << 16) | ((uint32_t)bi[w][3] << 24);
      } else {
        uint2 x;
        x.x = (uint32_t)bi[w][0] | ((uint32_t)bi[w][1] << 8) | ((uint32_t)bi[w][2] << 16) | ((uint32_t)bi[w][3] << 24);
        x.y = (uint32_t)bi[w][4] | ((uint32_t)bi[w][5] << 8) | ((uint32_t)bi[w][6] << 16) | ((uint32_t)bi[w][7] << 24);
        *reinterpret_cast<uint2*>(idx + o) = x;
      }
    }
  }
}

// The same operator for bf16 on PACKED KEYS (round 6).  The kernel above is bound by its compare / select arithmetic (~19 VALU operations per input
// element: 210 us for 565 MB = 2.7 TB/s).  After ReLU every value is a non-negative bf16, whose bit pattern orders like an unsigned integer, so
// (value << 16) | (15 - tap code) is a key whose integer maximum is the window's maximum WITH the first-maximum tie rule of the scan above (equal values:
// the smaller tap code wins): one v_lshl_or / v_and_or + one v_max_u32 per (element, window) instead of a compare and two selects, BatchNorm + ReLU on
// packed pairs (round to bf16, then v_pk_max_i16 with 0), the ReLU bits from the packed values.  Identical pooled values and argmax codes; the ReLU bit is
// (stored value != 0), which differs from (fp32 value > 0) only where a positive fp32 value rounds to a bf16 zero (below 2^-134).
static __device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {   // (hipcc turns min(x, 1) into a compare + select per half)
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ __launch_bounds__(256) void bn_relu_maxpool3_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                        bf16_t* __restrict__ p, uint8_t* __restrict__ idx, uint8_t* __restrict__ bits, int N, int H, int W, int C,
                                        int Ho, int Wo) {
  constexpr int V = 8;
  const int cv = C / V;
  const int Hb = Ho / 2, Wb = Wo / 2;
  const size_t total = (size_t)N * Hb * Wb * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cvi = (int)(i % cv);
    const int c0 = cvi * V;
    size_t t = i / cv;
    const int bw = (int)(t % Wb);
    t /= Wb;
    const int bh = (int)(t % Hb);
    const int n = (int)(t / Hb);
    float sc[V], sh[V];
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c0 + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(shift + c0 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[4 * q + e] = a[e];
        sh[4 * q + e] = b[e];
      }
    }
    uint32_t best[4][V];   // (value << 16) | (15 - code); every window has at least four taps, each key is >= 7: 0 is "none yet"
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < V; ++e) best[w][e] = 0u;
    const int ih0 = 4 * bh - 1, iw0 = 4 * bw - 1;  // top-left corner of the 5 x 5 patch
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int ih = ih0 + r;
      if (r == 0 && bh == 0) continue;   // (H = 4 Hb, W = 4 Wb: only the patch's first row / column can lie outside the image)
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int iw = iw0 + c;
        if (c == 0 && bw == 0) continue;
        const size_t pix = ((size_t)n * H + ih) * W + iw;
        const uint4 raw = *reinterpret_cast<const uint4*>(y + pix * C + c0);
        const uint32_t d[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = fmaf(__uint_as_float(d[j] << 16), sc[2 * j], sh[2 * j]);
          const float hi = fmaf(__uint_as_float(d[j] & 0xffff0000u), sc[2 * j + 1], sh[2 * j + 1]);
          typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          bf16x2 t2 = {(bf16_t)lo, (bf16_t)hi};                       // v_cvt_pk_bf16_f32 (RNE)
          s16x2 s2 = __builtin_bit_cast(s16x2, t2);
          s2 = __builtin_elementwise_max(s2, (s16x2){0, 0});          // ReLU on the packed pair: a negative bf16 (-0 included) is a negative int16
          pk[j] = __builtin_bit_cast(uint32_t, s2);
        }
        if (r >= 1 && c >= 1) {  // rows / columns 1..4 of the patch are owned by this block: bit e = (stored value e != 0)
          uint32_t t4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t4[j] = pk_min_u16(pk[j], 0x00010001u);   // 0 / 1 per half
          const uint32_t g = (t4[0] | (t4[1] << 2)) | ((t4[2] | (t4[3] << 2)) << 4);   // even elements at bits 0, 2, 4, 6; odd ones at 16, 18, 20, 22
          bits[pix * cv + cvi] = (uint8_t)((g & 0x55u) | ((g >> 15) & 0xaau));
        }
        // windows (dr, dc) of the block that contain patch element (r, c): kh = r - 2 dr, kw = c - 2 dc in 0..2
#pragma unroll
        for (int dr = 0; dr < 2; ++dr) {
          const int kh = r - 2 * dr;
          if (kh < 0 || kh > 2) continue;
#pragma unroll
          for (int dc = 0; dc < 2; ++dc) {
            const int kw = c - 2 * dc;
            if (kw < 0 || kw > 2) continue;
            const int w = 2 * dr + dc;
            const uint32_t code = 15u - (uint32_t)(kh * 3 + kw);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t klo = (pk[j] << 16) | code, khi = (pk[j] & 0xffff0000u) | code;
              best[w][2 * j] = __builtin_elementwise_max(best[w][2 * j], klo);
              best[w][2 * j + 1] = __builtin_elementwise_max(best[w][2 * j + 1], khi);
            }
          }
        }
      }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const size_t o = (((size_t)n * Ho + 2 * bh + (w >> 1)) * Wo + 2 * bw + (w & 1)) * C + c0;
      uint4 v;
      v.x = (best[w][0] >> 16) | (best[w][1] & 0xffff0000u);
      v.y = (best[w][2] >> 16) | (best[w][3] & 0xffff0000u);
      v.z = (best[w][4] >> 16) | (best[w][5] & 0xffff0000u);
      v.w = (best[w][6] >> 16) | (best[w][7] & 0xffff0000u);
      *reinterpret_cast<uint4*>(p + o) = v;
      uint2 x;
      x.x = (15u - (best[w][0] & 15u)) | ((15u - (best[w][1] & 15u)) << 8) | ((15u - (best[w][2] & 15u)) << 16) | ((15u - (best[w][3] & 15u)) << 24);
      x.y = (15u - (best[w][4] & 15u)) | ((15u - (best[w][5] & 15u)) << 8) | ((15u - (best[w][6] & 15u)) << 16) | ((15u - (best[w][7] & 15u)) << 24);
      *reinterpret_cast<uint2*>(idx + o) = x;
    }
  }
}

// Gather form (no atomics): one thread per 2 x 2 block of INPUT pixels and channel vector.  The four windows (a,b), (a,b+1), (a+1,b),
// (a+1,b+1) are the only ones that touch the block (an even row / column lies in one window, an odd one in two), so 4 loads of
// dy + argmax serve the 9 (pixel, window) pairs a thread per pixel fetched with 9 loads.  A pixel adds its windows in (oh, ow)
// order, as the per-pixel form did (bitwise the same result).
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx,
                                   int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int V = Vec16<T>::N;
  const int cv = C / V;
  const size_t total = (size_t)N * Ho * Wo * cv;  // H = 2 Ho, W = 2 Wo
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cv) * V;
    size_t t = i / cv;
    const int b = (int)(t % Wo);
    t /= Wo;
    const int a = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float g[4][V];
    uint8_t id[4][V];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int oh = a + (w >> 1), ow = b + (w & 1);
      const bool ok = oh < Ho && ow < Wo;
      const size_t o = (((size_t)n * Ho + (ok ? oh : a)) * Wo + (ok ? ow : b)) * C + c0;
      Vec16<T>::load(dy + o, g[w]);
      if constexpr (V == 4) {
        const uint32_t x = *reinterpret_cast<const uint32_t*>(idx + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) id[w][e] = ok ? (uint8_t)(x >> (8 * e)) : (uint8_t)255;
      } else {
        const uint2 x = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          id[w][e] = ok ? (uint8_t)(x.x >> (8 * e)) : (uint8_t)255;
          id[w][4 + e] = ok ? (uint8_t)(x.y >> (8 * e)) : (uint8_t)255;
        }
      }
    }
    // position (kh * 3 + kw) of block pixel (r, c) inside window w = 2 * dr + dc:  kh = r + 1 - 2 dr, kw = c + 1 - 2 dc
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dr = 0; dr <= r; ++dr)
#pragma unroll
          for (int dc = 0; dc <= c; ++dc) {
            const int pos = (r + 1 - 2 * dr) * 3 + (c + 1 - 2 * dc);
#pragma unroll
            for (int e = 0; e < V; ++e)
              if (id[2 * dr + dc][e] == pos) acc[e] += g[2 * dr + dc][e];
          }
        Vec16<T>::store(dx + (((size_t)n * H + 2 * a + r) * W + 2 * b + c) * C + c0, acc);
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// one workgroup per (image, slab of 8 channel vectors): 256 threads = 32 pixel lanes x 8 vectors of 16 bytes; every thread adds
// its pixels p = lane, lane + 32, ... in order, then a fixed-shape LDS tree over the 32 lanes (bitwise reproducible).  A single
// thread per (image, vector) walking all HW pixels was fine for the 7 x 7 head but took 1.5 ms on the 56 x 56 x 256 maps ECA pools.
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ pooled, int N, int HW, int C) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[32][8 * V + 1];
  const int slabs = C / (8 * V);
  const int n = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * 8 * V;
  const int cv = threadIdx.x & 7, r = threadIdx.x >> 3;
  float s[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s[e] = 0.f;
  for (int p = r; p < HW; p += 32) {
    float v[V];
    Vec16<T>::load(x + ((size_t)n * HW + p) * C + c0 + cv * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[r][cv * V + e] = s[e];
  __syncthreads();
  for (int st = 16; st > 0; st >>= 1) {
    if (r < st) {
#pragma unroll
      for (int e = 0; e < V; ++e) red[r][cv * V + e] += red[r + st][cv * V + e];
    }
    __syncthreads();
  }
  if (r == 0) {
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < V; ++e) pooled[(size_t)n * C + c0 + cv * V + e] = red[0][cv * V + e] * inv;
  }
}

template <typename T>
__global__ void gap_bwd_kernel(const float* __restrict__ dpooled, T* __restrict__ dx, int N, int HW, int C) {
  constexpr int V = Vec16<T>::N;
  const int cv = C / V;
  const size_t total = (size_t)N * HW * cv;
  const float inv = 1.f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cv) * V;
    const int n = (int)(i / ((size_t)cv * HW));
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = dpooled[(size_t)n * C + c0 + e] * inv;
    Vec16<T>::store(dx + i * V, v);
  }
}

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, sh[w]);
  return t;
}

// one workgroup per sample
__global__ __launch_bounds__(256) void ce_row_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                     float smoothing, float gscale, float* __restrict__ row_loss,
                                                     float* __restrict__ dlogits, int N, int C) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  const float* z = logits + (size_t)n * C;
  const float* y = target + (size_t)n * C;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, z[c]);
  mx = block_max(mx, sh);
  float se = 0.f, sy = 0.f, syz = 0.f, sz = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float zc = z[c], yc = y[c];
    se += expf(zc - mx);
    sy += yc;
    syz += yc * zc;
    sz += zc;
  }
  se = block_sum(se, sh);
  sy = block_sum(sy, sh);
  syz = block_sum(syz, sh);
  sz = block_sum(sz, sh);
  const float lse = mx + logf(se);
  if (threadIdx.x == 0) {
    const float nll = -(syz - lse * sy);
    const float uni = -(sz / (float)C - lse);
    row_loss[n] = (1.f - smoothing) * nll + smoothing * uni;
  }
  if (dlogits) {
    const float k = gscale / (float)N;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float pc = expf(z[c] - lse);
      dlogits[(size_t)n * C + c] = k * ((1.f - smoothing) * (pc * sy - y[c]) + smoothing * (pc - 1.f / (float)C));
    }
  }
}

__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ row_loss, float* __restrict__ loss, int N) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += row_loss[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *loss = s / (float)N;
}

// ---------------------------------------------------------------------------------------------------
// EMA: also ema = ema + (1 - decay) * (p_new - ema) — the recipe's ModelEma (train.py:111-112) in the pass that has p in registers anyway
template <bool EMA>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ ema,
                                                  size_t n4, size_t n, float lr, float mom, float wd, float gscale, float ema_w) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 ev;
    if constexpr (EMA) ev = reinterpret_cast<f32x4*>(ema)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gv[e] * gscale + wd * pv[e];
      mv[e] = mom * mv[e] + ge;
      pv[e] = pv[e] - lr * mv[e];
      if constexpr (EMA) ev[e] = ev[e] + ema_w * (pv[e] - ev[e]);
    }
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
    if constexpr (EMA) reinterpret_cast<f32x4*>(ema)[i] = ev;
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float ge = g[i] * gscale + wd * p[i];
    const float me = mom * m[i] + ge;
    m[i] = me;
    const float pn = p[i] - lr * me;
    p[i] = pn;
    if constexpr (EMA) ema[i] = ema[i] + ema_w * (pn - ema[i]);
  }
}

__global__ void bias_slice_kernel(const float* __restrict__ tmp, int ld, const float* __restrict__ bias,
                                  float* __restrict__ out, int N, int O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * O) return;
  const int o = i % O, n = i / O;
  out[i] = tmp[(size_t)n * ld + o] + bias[o];
}

__global__ void pad_dlogits_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld, int N, int O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * ld) return;
  const int o = i % ld, n = i / ld;
  dst[i] = o < O ? src[(size_t)n * O + o] : 0.f;
}

// dbias[o] = sum_n src[n][o]: 64 outputs per workgroup, 16 row lanes each adding every 16th row in order, then a fixed LDS tree
// (one thread per output walking all N rows took 45 us at N = 256 on the critical path of the backward)
__global__ __launch_bounds__(1024) void dbias_kernel(const float* __restrict__ src, float* __restrict__ dbias, float beta, int N, int O) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + c;
  float s = 0.f;
  if (o < O)
    for (int n = r; n < N; n += 16) s += src[(size_t)n * O + o];
  red[r][c] = s;
  __syncthreads();
  for (int st = 8; st > 0; st >>= 1) {
    if (r < st) red[r][c] += red[r + st][c];
    __syncthreads();
  }
  if (r == 0 && o < O) dbias[o] = (beta != 0.f ? beta * dbias[o] : 0.f) + red[0][c];
}

int grid_for(size_t total, int cap = 8192) {
  size_t b = (total + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

#define DISPATCH_T(dtype, KERNEL, grid, ...)                                                    \
  do {                                                                                          \
    if ((dtype) == MI355_F32)                                                                   \
      hipLaunchKernelGGL(KERNEL<float>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);              \
    else if ((dtype) == MI355_BF16)                                                             \
      hipLaunchKernelGGL(KERNEL<bf16_t>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);             \
    else {                                                                                      \
      set_error("bad dtype %d", (dtype));                                                       \
      return MI355_E_ARG;                                                                       \
    }                                                                                           \
    MI355_LAUNCH_CHECK();                                                                       \
  } while (0)

}  // namespace

int launch_stem_ingest(int dtype, const float* x, void* xpad, int N, int H, int W, hipStream_t s) {
  const int Hp = stem_hp(H), Wp = stem_wp(W);
  const int grid = grid_for((size_t)N * H * W);
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(stem_ingest_kernel<float>, dim3(grid), dim3(256), 0, s, x, (float*)xpad, N, H, W, Hp, Wp);
  else
    hipLaunchKernelGGL(stem_ingest_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, (bf16_t*)xpad, N, H, W, Hp, Wp);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_maxpool_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, hipStream_t s) {
  MI355_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "maxpool: H=%d W=%d C=%d", H, W, C);
  const int Ho = H / 2, Wo = W / 2;
  const int V = 16 / (int)dtype_size(dtype);
  const int grid = grid_for((size_t)N * Ho * Wo * (C / V));
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, idx, N, H, W,
                       C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N,
                       H, W, C, Ho, Wo);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_relu_maxpool(int dtype, const void* y, const float* scale, const float* shift, void* p, uint8_t* idx,
                           uint8_t* bits, int N, int H, int W, int C, hipStream_t s) {
  MI355_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "bn_relu_maxpool: H=%d W=%d C=%d", H, W, C);
  const int Ho = H / 2, Wo = W / 2;
  const int V = 16 / (int)dtype_size(dtype);
  if (Ho % 2 == 0 && Wo % 2 == 0) {  // 2 x 2 output blocks (every size the network is run at: H, W multiples of 32)
    const int grid = grid_for((size_t)N * (Ho / 2) * (Wo / 2) * (C / V));
    if (dtype == MI355_F32)
      hipLaunchKernelGGL(bn_relu_maxpool2_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)y, scale, shift, (float*)p, idx, bits, N, H, W,
                         C, Ho, Wo);
    else if (knobs().pool_keys)
      hipLaunchKernelGGL(bn_relu_maxpool3_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift, (bf16_t*)p, idx, bits, N, H, W, C, Ho, Wo);
    else
      hipLaunchKernelGGL(bn_relu_maxpool2_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift, (bf16_t*)p, idx, bits, N,
                         H, W, C, Ho, Wo);
    MI355_LAUNCH_CHECK();
    return 0;
  }
  const int grid = grid_for((size_t)N * Ho * Wo * (C / V));
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)y, scale, shift, (float*)p,
                       idx, bits, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                       (bf16_t*)p, idx, bits, N, H, W, C, Ho, Wo);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_maxpool_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                       hipStream_t s) {
  MI355_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "maxpool: H=%d W=%d C=%d", H, W, C);
  const int Ho = H / 2, Wo = W / 2;
  const int V = 16 / (int)dtype_size(dtype);
  const int grid = grid_for((size_t)N * Ho * Wo * (C / V));
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, N, H,
                       W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, N,
                       H, W, C, Ho, Wo);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_gap_fwd(int dtype, const void* x, float* pooled, int N, int HW, int C, hipStream_t s) {
  MI355_ARG(C % 8 == 0, "gap: C=%d", C);
  const int V = 16 / (int)dtype_size(dtype);
  MI355_ARG(C % (8 * V) == 0, "gap: C=%d must be a multiple of %d", C, 8 * V);
  const int grid = N * (C / (8 * V));
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, pooled, N, HW, C);
  else
    hipLaunchKernelGGL(gap_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, pooled, N, HW, C);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_gap_bwd(int dtype, const float* dpooled, void* dx, int N, int HW, int C, hipStream_t s) {
  MI355_ARG(C % 8 == 0, "gap: C=%d", C);
  const int V = 16 / (int)dtype_size(dtype);
  const int grid = grid_for((size_t)N * HW * (C / V));
  if (dtype == MI355_F32)
    hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, dpooled, (float*)dx, N, HW, C);
  else
    hipLaunchKernelGGL(gap_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, dpooled, (bf16_t*)dx, N, HW, C);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_ce(const float* logits, const float* target, float smoothing, float grad_scale, float* loss,
              float* row_loss, float* dlogits, int N, int C, hipStream_t s) {
  MI355_ARG(logits && target && loss && row_loss, "ce: null pointer");
  hipLaunchKernelGGL(ce_row_kernel, dim3(N), dim3(256), 0, s, logits, target, smoothing, grad_scale, row_loss, dlogits,
                     N, C);
  MI355_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, s, row_loss, loss, N);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_sgd(float* p, const float* g, float* m, size_t n, float lr, float mom, float wd, float gscale,
               hipStream_t s, float* ema, float ema_decay) {
  MI355_ARG(p && g && m, "sgd: null pointer");
  MI355_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)ema % 16 == 0),
            "sgd: pointers must be 16-byte aligned");
  MI355_ARG(!ema || (ema_decay >= 0.f && ema_decay <= 1.f), "sgd: ema_decay=%g outside [0, 1]", (double)ema_decay);
  const size_t n4 = n / 4;
  if (ema) hipLaunchKernelGGL(sgd_kernel<true>, dim3(grid_for(n4, 4096)), dim3(256), 0, s, p, g, m, ema, n4, n, lr, mom, wd, gscale, 1.f - ema_decay);
  else hipLaunchKernelGGL(sgd_kernel<false>, dim3(grid_for(n4, 4096)), dim3(256), 0, s, p, g, m, ema, n4, n, lr, mom, wd, gscale, 0.f);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bias_slice(const float* tmp, int ld, const float* bias, float* out, int N, int O, hipStream_t s) {
  hipLaunchKernelGGL(bias_slice_kernel, dim3(cdiv(N * O, 256)), dim3(256), 0, s, tmp, ld, bias, out, N, O);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_pad_dlogits(const float* src, float* dst, int ld, float* dbias, float beta, int N, int O, hipStream_t s) {
  hipLaunchKernelGGL(pad_dlogits_kernel, dim3(cdiv(N * ld, 256)), dim3(256), 0, s, src, dst, ld, N, O);
  MI355_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(dbias_kernel, dim3(cdiv(O, 64)), dim3(1024), 0, s, src, dbias, beta, N, O);
    MI355_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace mi355
