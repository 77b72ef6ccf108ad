// ingest.hip — real-image ingest for gfx950 (MI355X): decoded u8 RGB crops -> the loader's NCHW fp32 batch, on the device.
//
// Replaces the GPU half of the reference's DALI pipelines (the part after the JPEG decoder):
//   train  sota_imagenet/dali_dataloader.py:69-78  random crop (chosen by the host at decode time, like DALI's
//          image_random_crop) -> fn.resize(size = S, INTERP_TRIANGULAR | INTERP_CUBIC by coin :78-83) ->
//          gaussian_blur(window 11, sigma) :85-87 -> color_twist :89-98 -> hsv(saturation = coin) :100-102 ->
//          erase(fill = mean) :104-114 -> crop_mirror_normalize(mirror = coin, mean 127.5, std 51 (:27-29), FLOAT, NCHW) :116-125
//   val    :144-157  resize_shorter = ceil((S * 1.14 + 8) // 16 * 16), centre crop S x S, normalise, NCHW
// "Resize the source rectangle to rh x rw, cut the S x S window at (oy, ox), augment, optionally mirror, normalise."  One
// launch per batch reads crops of DIFFERENT sizes from one packed byte buffer through a per-sample descriptor table (two
// launches when some sample of the batch is blurred: the 11 x 11 window needs its neighbours' resized values, which go
// through an fp32 scratch image) — no per-image launches.
// Filters (the reference's DALI build is unpinned; these are the textbook separable kernels with Pillow's border law, which is
// what pins the oracle — tests/test_image_loader_host.py):
//   scale = in / out, support = R * max(scale, 1) (R = 1 triangle, 2 cubic), centre = (o + 0.5) * scale,
//   taps x in [max(0, int(centre - support + 0.5)), min(in, int(centre + support + 0.5))), weight f((x + 0.5 - centre) / max(scale, 1)),
//   f = triangle 1 - |t|, or the a = -0.5 cubic convolution kernel; normalised by their sum (borders renormalise, no padding).
// HBM-bound by design: source bytes come from L2 (a batch of crops is tens of MB), the fp32 output is written once
// (N*3*S*S*4 bytes: 154 MB at batch 256, 224 px).
#include "common.h"

namespace mi355 {
namespace {

struct Crop {  // = mi355_crop (include/mi355rn.h)
  unsigned long long offset;  // first byte of the crop's tightly packed HWC RGB rows in the packed buffer
  int h, w;                   // source rectangle
  int rh, rw;                 // size it is resized to
  int oy, ox;                 // top-left corner of the S x S output window inside the resized image
  int mirror;                 // 1: horizontal flip of the window (applied last, like crop_mirror_normalize)
  int filter;                 // 0 triangular, 1 cubic
};
static_assert(sizeof(Crop) == 40, "descriptor layout is part of the C-ABI");

struct Augment {  // = mi355_augment
  float color[12];  // 3 x 4 row-major: v'_c = sum_k color[4c+k] * v_k + color[4c+3] on 0..255 values, then clamp to [0, 255]
  float blur_sigma; // 0: none; else gaussian over the resized image, window 11, reflect-101 border
  int gray;         // 1: replace by luma (0.299, 0.587, 0.114) after the colour step
  int nbox;         // erased rectangles (<= 4), filled with `mean` (= 0 after normalisation)
  int pad;
  int box[4][4];    // y0, x0, y1, x1 (half open) in window coordinates BEFORE the mirror
};
static_assert(sizeof(Augment) == 128, "descriptor layout is part of the C-ABI");

struct Axis {
  float centre, inv_scale;
  int lo, hi;
};
__device__ __forceinline__ Axis axis_of(int o, int in, int out, int filter) {
  const float scale = (float)in / (float)out;
  const float fs = scale > 1.f ? scale : 1.f;
  const float support = (filter ? 2.f : 1.f) * fs;
  Axis a;
  a.centre = ((float)o + 0.5f) * scale;
  a.inv_scale = 1.f / fs;
  a.lo = (int)(a.centre - support + 0.5f);
  a.hi = (int)(a.centre + support + 0.5f);
  a.lo = a.lo < 0 ? 0 : a.lo;
  a.hi = a.hi > in ? in : a.hi;
  return a;
}
__device__ __forceinline__ float tap(int x, const Axis& a, int filter) {
  const float t = fabsf(((float)x + 0.5f - a.centre) * a.inv_scale);
  if (!filter) return t < 1.f ? 1.f - t : 0.f;
  if (t < 1.f) return (1.5f * t - 2.5f) * t * t + 1.f;                 // a = -0.5: (a + 2) t^3 - (a + 3) t^2 + 1
  if (t < 2.f) return ((-0.5f * t + 2.5f) * t - 4.f) * t + 2.f;        // a t^3 - 5a t^2 + 8a t - 4a
  return 0.f;
}

// resized value of window pixel (i, jj) (jj = column before the mirror), 0..255 scale
__device__ __forceinline__ void resample(const unsigned char* __restrict__ packed, const Crop& c, int i, int jj, float& r, float& g, float& b) {
  const Axis ay = axis_of(c.oy + i, c.h, c.rh, c.filter);
  const Axis ax = axis_of(c.ox + jj, c.w, c.rw, c.filter);
  const unsigned char* src = packed + c.offset;
  float wsum = 0.f;
  r = g = b = 0.f;
  for (int y = ay.lo; y < ay.hi; ++y) {
    const float wy = tap(y, ay, c.filter);
    const unsigned char* row = src + (size_t)y * c.w * 3;
    float rr = 0.f, gg = 0.f, bb = 0.f, ws = 0.f;
    for (int x = ax.lo; x < ax.hi; ++x) {
      const float wx = tap(x, ax, c.filter);
      rr += wx * (float)row[x * 3 + 0];
      gg += wx * (float)row[x * 3 + 1];
      bb += wx * (float)row[x * 3 + 2];
      ws += wx;
    }
    r += wy * rr;
    g += wy * gg;
    b += wy * bb;
    wsum += wy * ws;
  }
  const float inv = 1.f / wsum;
  r *= inv;
  g *= inv;
  b *= inv;
  if (c.filter) {  // the cubic overshoots: an 8-bit image saturates (the reference's operators hand u8 images to each other)
    r = fminf(fmaxf(r, 0.f), 255.f);
    g = fminf(fmaxf(g, 0.f), 255.f);
    b = fminf(fmaxf(b, 0.f), 255.f);
  }
}

__device__ __forceinline__ void augment_px(const Augment& a, int i, int jj, float mean, float& r, float& g, float& b) {
  const float r1 = a.color[0] * r + a.color[1] * g + a.color[2] * b + a.color[3];
  const float g1 = a.color[4] * r + a.color[5] * g + a.color[6] * b + a.color[7];
  const float b1 = a.color[8] * r + a.color[9] * g + a.color[10] * b + a.color[11];
  r = fminf(fmaxf(r1, 0.f), 255.f);
  g = fminf(fmaxf(g1, 0.f), 255.f);
  b = fminf(fmaxf(b1, 0.f), 255.f);
  if (a.gray) r = g = b = 0.299f * r + 0.587f * g + 0.114f * b;
  for (int k = 0; k < a.nbox; ++k)
    if (i >= a.box[k][0] && i < a.box[k][2] && jj >= a.box[k][1] && jj < a.box[k][3]) r = g = b = mean;
}

// MODE 0: resize (+ augment when aug != nullptr) + mirror + normalise -> out.  MODE 1: resize only -> scratch (0..255, unmirrored)
template <int MODE>
__global__ __launch_bounds__(256) void ingest_kernel(const unsigned char* __restrict__ packed, const Crop* __restrict__ crops, const Augment* __restrict__ aug,
                                                     int S, float mean, float inv_std, float* __restrict__ out) {
  const int n = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= S * S) return;
  const Crop c = crops[n];
  const int i = pix / S, j = pix - i * S;
  const int jj = (MODE == 0 && c.mirror) ? S - 1 - j : j;
  float r, g, b;
  resample(packed, c, i, jj, r, g, b);
  const size_t plane = (size_t)S * S;
  float* o = out + (size_t)n * 3 * plane + pix;
  if (MODE == 1) {
    o[0] = r;
    o[plane] = g;
    o[2 * plane] = b;
    return;
  }
  if (aug) augment_px(aug[n], i, jj, mean, r, g, b);
  o[0] = (r - mean) * inv_std;
  o[plane] = (g - mean) * inv_std;
  o[2 * plane] = (b - mean) * inv_std;
}

// second pass of a blurred batch: 11 x 11 gaussian (separable weights, reflect-101) over the scratch image, then the rest
__global__ __launch_bounds__(256) void blur_augment_kernel(const float* __restrict__ scratch, const Crop* __restrict__ crops, const Augment* __restrict__ aug, int S,
                                                           float mean, float inv_std, float* __restrict__ out) {
  const int n = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= S * S) return;
  const Augment a = aug[n];
  const int i = pix / S, j = pix - i * S;
  const int jj = crops[n].mirror ? S - 1 - j : j;
  const size_t plane = (size_t)S * S;
  const float* src = scratch + (size_t)n * 3 * plane;
  float r, g, b;
  if (a.blur_sigma > 0.f) {
    float w[6], wsum = 0.f;
    const float k = -0.5f / (a.blur_sigma * a.blur_sigma);
    for (int d = 0; d <= 5; ++d) {
      w[d] = __expf(k * (float)(d * d));
      wsum += d ? 2.f * w[d] : w[d];
    }
    const float inv = 1.f / wsum;
    r = g = b = 0.f;
    for (int dy = -5; dy <= 5; ++dy) {
      int y = i + dy;
      y = y < 0 ? -y : (y >= S ? 2 * S - 2 - y : y);
      y = y < 0 ? 0 : (y >= S ? S - 1 : y);  // (windows wider than the image: clamp)
      const float wy = w[dy < 0 ? -dy : dy] * inv;
      float rr = 0.f, gg = 0.f, bb = 0.f;
      for (int dx = -5; dx <= 5; ++dx) {
        int x = jj + dx;
        x = x < 0 ? -x : (x >= S ? 2 * S - 2 - x : x);
        x = x < 0 ? 0 : (x >= S ? S - 1 : x);
        const float wx = w[dx < 0 ? -dx : dx] * inv;
        const size_t o = (size_t)y * S + x;
        rr += wx * src[o];
        gg += wx * src[plane + o];
        bb += wx * src[2 * plane + o];
      }
      r += wy * rr;
      g += wy * gg;
      b += wy * bb;
    }
  } else {
    const size_t o = (size_t)i * S + jj;
    r = src[o];
    g = src[plane + o];
    b = src[2 * plane + o];
  }
  augment_px(a, i, jj, mean, r, g, b);
  float* o = out + (size_t)n * 3 * plane + pix;
  o[0] = (r - mean) * inv_std;
  o[plane] = (g - mean) * inv_std;
  o[2 * plane] = (b - mean) * inv_std;
}

int check_crops(const mi355_crop* crops_host, size_t packed_bytes, int N, int S) {
  // the descriptors index raw memory: check every one on the host before the launch (a bad table is a caller bug that must not
  // become an out-of-bounds read on the device)
  for (int n = 0; n < N; ++n) {
    const mi355_crop& c = crops_host[n];
    MI355_ARG(c.h > 0 && c.w > 0 && c.rh > 0 && c.rw > 0, "ingest_u8: sample %d has an empty rectangle (%d x %d -> %d x %d)", n, c.h, c.w, c.rh, c.rw);
    MI355_ARG(c.oy >= 0 && c.ox >= 0 && c.oy + S <= c.rh && c.ox + S <= c.rw, "ingest_u8: sample %d: the %d px window at (%d, %d) leaves the %d x %d image", n, S,
              c.oy, c.ox, c.rh, c.rw);
    MI355_ARG(c.offset + (unsigned long long)c.h * c.w * 3 <= packed_bytes, "ingest_u8: sample %d ends past the packed buffer (%zu bytes)", n, packed_bytes);
    MI355_ARG(c.mirror == 0 || c.mirror == 1, "ingest_u8: sample %d mirror=%d", n, c.mirror);
    MI355_ARG(c.filter == 0 || c.filter == 1, "ingest_u8: sample %d filter=%d (0 triangular, 1 cubic)", n, c.filter);
  }
  return 0;
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_ingest_u8(const unsigned char* packed, size_t packed_bytes, const mi355_crop* crops_host, const mi355_crop* crops_dev, int N, int S,
                               float mean, float std, float* out_nchw, void* stream) {
  MI355_ARG(packed && crops_host && crops_dev && out_nchw && N > 0 && S > 0 && std > 0.f, "ingest_u8: N=%d S=%d", N, S);
  MI355_TRY(check_crops(crops_host, packed_bytes, N, S));
  hipLaunchKernelGGL(ingest_kernel<0>, dim3((unsigned)((S * S + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, packed,
                     reinterpret_cast<const Crop*>(crops_dev), (const Augment*)nullptr, S, mean, 1.f / std, out_nchw);
  MI355_LAUNCH_CHECK();
  return 0;
}

extern "C" int mi355_ingest_u8_aug(const unsigned char* packed, size_t packed_bytes, const mi355_crop* crops_host, const mi355_crop* crops_dev,
                                   const mi355_augment* aug_host, const mi355_augment* aug_dev, int N, int S, float mean, float std, float* scratch,
                                   float* out_nchw, void* stream) {
  MI355_ARG(packed && crops_host && crops_dev && aug_host && aug_dev && out_nchw && N > 0 && S > 0 && std > 0.f, "ingest_u8_aug: N=%d S=%d", N, S);
  MI355_TRY(check_crops(crops_host, packed_bytes, N, S));
  bool blur = false;
  for (int n = 0; n < N; ++n) {
    const mi355_augment& a = aug_host[n];
    MI355_ARG(a.nbox >= 0 && a.nbox <= 4 && (a.gray == 0 || a.gray == 1) && a.blur_sigma >= 0.f, "ingest_u8_aug: sample %d: nbox=%d gray=%d sigma=%f", n, a.nbox, a.gray,
              a.blur_sigma);
    blur = blur || a.blur_sigma > 0.f;
  }
  MI355_ARG(!blur || scratch, "ingest_u8_aug: a blurred sample needs the N*3*S*S float scratch image");
  const dim3 grid((unsigned)((S * S + 255) / 256), (unsigned)N);
  const Crop* cd = reinterpret_cast<const Crop*>(crops_dev);
  const Augment* ad = reinterpret_cast<const Augment*>(aug_dev);
  if (!blur) {
    hipLaunchKernelGGL(ingest_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, packed, cd, ad, S, mean, 1.f / std, out_nchw);
  } else {
    hipLaunchKernelGGL(ingest_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, packed, cd, ad, S, mean, 1.f / std, scratch);
    hipLaunchKernelGGL(blur_augment_kernel, grid, dim3(256), 0, (hipStream_t)stream, scratch, cd, ad, S, mean, 1.f / std, out_nchw);
  }
  MI355_LAUNCH_CHECK();
  return 0;
}
