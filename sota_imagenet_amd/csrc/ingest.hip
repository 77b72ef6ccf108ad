// ingest.hip — real-image ingest for gfx950 (MI355X): decoded u8 RGB crops -> the loader's NCHW fp32 batch, on the device.
//
// Replaces the GPU half of the reference's DALI pipelines (the part after the JPEG decoder):
//   train  sota_imagenet/dali_dataloader.py:69-78  random crop (chosen by the host at decode time, like DALI's
//          image_random_crop) -> fn.resize(size = S, INTERP_TRIANGULAR) -> :111-120 crop_mirror_normalize(mirror = coin,
//          mean 127.5, std 51 (:27-29), FLOAT, NCHW)
//   val    :144-157  resize_shorter = ceil((S * 1.14 + 8) // 16 * 16), centre crop S x S, normalise, NCHW
// Both are "resize the source rectangle to rh x rw with a triangular (antialiased bilinear) filter, cut the S x S window at
// (oy, ox), optionally mirror it, normalise".  One launch per batch reads the crops of DIFFERENT sizes from one packed byte
// buffer through a per-sample descriptor table — no per-image launches, no intermediate resized image in HBM.
// Filter definition (the reference's DALI build is unpinned; this is the textbook separable triangle, the same law Pillow's
// BILINEAR resize follows, which pins the oracle: tests/test_image_loader_host.py):
//   scale = in / out, support = max(scale, 1), centre = (o + 0.5) * scale,
//   taps x in [max(0, int(centre - support + 0.5)), min(in, int(centre + support + 0.5))), weight max(0, 1 - |x + 0.5 - centre| / support),
//   normalised by their sum (so image borders renormalise instead of padding).
// HBM-bound by design: the source bytes are read (2*support+1)^2 times but from L2 (a batch of crops is tens of MB), the
// fp32 output is written once (N*3*S*S*4 bytes: 154 MB at batch 256, 224 px).
#include "common.h"

namespace mi355 {
namespace {

struct Crop {  // = mi355_crop (include/mi355rn.h)
  unsigned long long offset;  // first byte of the crop's tightly packed HWC RGB rows in the packed buffer
  int h, w;                   // source rectangle
  int rh, rw;                 // size it is resized to
  int oy, ox;                 // top-left corner of the S x S output window inside the resized image
  int mirror;                 // 1: horizontal flip of the window
  int pad;
};
static_assert(sizeof(Crop) == 40, "descriptor layout is part of the C-ABI");

struct Axis {
  float centre, inv_support;
  int lo, hi;
};
__device__ __forceinline__ Axis axis_of(int o, int in, int out) {
  const float scale = (float)in / (float)out;
  const float support = scale > 1.f ? scale : 1.f;
  Axis a;
  a.centre = ((float)o + 0.5f) * scale;
  a.inv_support = 1.f / support;
  a.lo = (int)(a.centre - support + 0.5f);
  a.hi = (int)(a.centre + support + 0.5f);
  a.lo = a.lo < 0 ? 0 : a.lo;
  a.hi = a.hi > in ? in : a.hi;
  return a;
}
__device__ __forceinline__ float tri(int x, const Axis& a) {
  const float t = fabsf(((float)x + 0.5f - a.centre) * a.inv_support);
  return t < 1.f ? 1.f - t : 0.f;
}

__global__ __launch_bounds__(256) void ingest_kernel(const unsigned char* __restrict__ packed, const Crop* __restrict__ crops, int S, float mean,
                                                     float inv_std, float* __restrict__ out) {
  const int n = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= S * S) return;
  const Crop c = crops[n];
  const int i = pix / S, j = pix - i * S;
  const Axis ay = axis_of(c.oy + i, c.h, c.rh);
  const Axis ax = axis_of(c.ox + (c.mirror ? S - 1 - j : j), c.w, c.rw);
  const unsigned char* src = packed + c.offset;
  float r = 0.f, g = 0.f, b = 0.f, wsum = 0.f;
  for (int y = ay.lo; y < ay.hi; ++y) {
    const float wy = tri(y, ay);
    const unsigned char* row = src + (size_t)y * c.w * 3;
    float rr = 0.f, gg = 0.f, bb = 0.f, ws = 0.f;
    for (int x = ax.lo; x < ax.hi; ++x) {
      const float wx = tri(x, ax);
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
  const size_t plane = (size_t)S * S;
  float* o = out + (size_t)n * 3 * plane + pix;
  o[0] = (r * inv - mean) * inv_std;
  o[plane] = (g * inv - mean) * inv_std;
  o[2 * plane] = (b * inv - mean) * inv_std;
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_ingest_u8(const unsigned char* packed, size_t packed_bytes, const mi355_crop* crops_host, const mi355_crop* crops_dev, int N, int S,
                               float mean, float std, float* out_nchw, void* stream) {
  MI355_ARG(packed && crops_host && crops_dev && out_nchw && N > 0 && S > 0 && std > 0.f, "ingest_u8: N=%d S=%d", N, S);
  // the descriptors index raw memory: check every one on the host before the launch (a bad table is a caller bug that must not
  // become an out-of-bounds read on the device)
  for (int n = 0; n < N; ++n) {
    const mi355_crop& c = crops_host[n];
    MI355_ARG(c.h > 0 && c.w > 0 && c.rh > 0 && c.rw > 0, "ingest_u8: sample %d has an empty rectangle (%d x %d -> %d x %d)", n, c.h, c.w, c.rh, c.rw);
    MI355_ARG(c.oy >= 0 && c.ox >= 0 && c.oy + S <= c.rh && c.ox + S <= c.rw, "ingest_u8: sample %d: the %d px window at (%d, %d) leaves the %d x %d image", n, S,
              c.oy, c.ox, c.rh, c.rw);
    MI355_ARG(c.offset + (unsigned long long)c.h * c.w * 3 <= packed_bytes, "ingest_u8: sample %d ends past the packed buffer (%zu bytes)", n, packed_bytes);
    MI355_ARG(c.mirror == 0 || c.mirror == 1, "ingest_u8: sample %d mirror=%d", n, c.mirror);
  }
  hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)((S * S + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, packed,
                     reinterpret_cast<const Crop*>(crops_dev), S, mean, 1.f / std, out_nchw);
  MI355_LAUNCH_CHECK();
  return 0;
}
