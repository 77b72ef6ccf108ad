// mix.hip — Mixup / CutMix of a batch with the previous batch, on the device, for gfx950 (MI355X).
//
// Replaces the tensor work of sota_imagenet/callbacks.py:232-247 (CutmixMixup: a coin between self.cutmix(*input) and
// self.mixup(*input), Beta(alpha, alpha) mixing weights; the un-vendored pytorch_tools Cutmix / Mixup bases mix with the
// PREVIOUS batch under a random permutation — SURVEY.md Appendix C).  Two launches per batch, nothing returns to the host:
//   mix_sample_kernel  ONE workgroup draws the batch's decisions from a counter-based generator (seed, batch counter):
//                      apply-or-not, CutMix-or-Mixup, lambda ~ Beta(alpha, alpha) (Marsaglia-Tsang gammas), the CutMix
//                      box, and a permutation of the batch (rank of per-sample random keys) -> a small parameter block
//   mix_apply_kernel   HBM-bound, float4: out <- mix(data, prev[perm]); prev' <- data (the unmixed batch, for the next
//                      step; prev is double-buffered because sample n's old row is read by some other sample);
//                      the soft targets [N, classes] are mixed with the same weights.
#include "common.h"

namespace mi355 {
namespace {

struct MixParams {  // the device-resident parameter block (mi355_mix_params_bytes): header + perm[N]
  int mode;         // 0 none, 1 mixup, 2 cutmix
  float lam;        // mixup: weight of the current batch; cutmix: the sampled lambda (box side = sqrt(lambda))
  int y1, y2, x1, x2;
  float lam_real;   // cutmix: box area / image area = weight of the previous batch's target
  int pad;
  int perm[1];
};

__device__ __forceinline__ unsigned long long splitmix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
struct Rng {
  unsigned long long key, n;
  __device__ float uniform() {  // (0, 1)
    const unsigned long long r = splitmix(key + 0xD1B54A32D192ED03ull * (++n));
    return ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
  }
  __device__ float normal() {  // Box-Muller
    const float u = uniform(), v = uniform();
    return sqrtf(-2.0f * logf(u)) * cosf(6.28318530718f * v);
  }
  __device__ float gamma(float a) {  // Marsaglia-Tsang; a < 1 through Gamma(a + 1) * U^(1/a)
    const float boost = a < 1.0f ? powf(uniform(), 1.0f / a) : 1.0f;
    if (a < 1.0f) a += 1.0f;
    const float d = a - 1.0f / 3.0f, c = rsqrtf(9.0f * d);
    for (int it = 0; it < 64; ++it) {
      const float x = normal(), t = 1.0f + c * x;
      if (t <= 0.0f) continue;
      const float v = t * t * t, u = uniform();
      if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) return d * v * boost;
    }
    return d * boost;  // (never in practice: the acceptance rate is > 95 %)
  }
  __device__ float beta(float a) {
    const float x = gamma(a), y = gamma(a);
    return x / (x + y);
  }
};

__global__ __launch_bounds__(1024) void mix_sample_kernel(MixParams* p, unsigned long long seed, unsigned long long counter, int N, int H,
                                                         int W, float cutmix_alpha, float mixup_alpha, float prob, int allow) {
  __shared__ unsigned long long keys[1024];
  const unsigned long long base = splitmix(seed ^ splitmix(counter));
  const int tid = threadIdx.x;
  if (tid == 0) {
    Rng r{base, 0};
    const bool apply = !(r.uniform() > prob);  // `if np.random.rand() > self.prob: return` of the bases
    bool cut = r.uniform() > 0.5f;             // callbacks.py:242
    if (allow == 1) cut = false;
    if (allow == 2) cut = true;
    int mode = 0, y1 = 0, y2 = 0, x1 = 0, x2 = 0;
    float lam = 1.0f, lam_real = 0.0f;
    if (apply && allow != 0) {
      lam = r.beta(cut ? cutmix_alpha : mixup_alpha);
      const float ucy = r.uniform(), ucx = r.uniform();
      if (cut) {
        mode = 2;
        const float l = fminf(lam, 1.0f - lam);
        const int bh = (int)((float)H * sqrtf(l)), bw = (int)((float)W * sqrtf(l));
        const int cy = min((int)(ucy * (float)H), H - 1), cx = min((int)(ucx * (float)W), W - 1);
        y1 = max(cy - bh / 2, 0); y2 = min(cy + bh / 2, H);
        x1 = max(cx - bw / 2, 0); x2 = min(cx + bw / 2, W);
        lam_real = (float)((y2 - y1) * (x2 - x1)) / (float)(H * W);
      } else {
        mode = 1;
      }
    }
    p->mode = mode; p->lam = lam; p->y1 = y1; p->y2 = y2; p->x1 = x1; p->x2 = x2; p->lam_real = lam_real; p->pad = 0;
  }
  // permutation: sample i goes to the rank of its key (ties broken by index) — deterministic, no atomics
  for (int i = tid; i < N; i += blockDim.x) keys[i] = splitmix(base + 0x632BE59BD9B4E019ull * (unsigned long long)(i + 1));
  __syncthreads();
  for (int i = tid; i < N; i += blockDim.x) {
    const unsigned long long k = keys[i];
    int rank = 0;
    for (int j = 0; j < N; ++j) rank += (keys[j] < k || (keys[j] == k && j < i)) ? 1 : 0;
    p->perm[i] = rank;
  }
}

// one thread = one float4 of the image tensor [N][3][H][W] (W % 4 == 0) or of the target matrix [N][C] (C % 4 == 0)
__global__ __launch_bounds__(256) void mix_apply_kernel(const float* data, float* out, const float* __restrict__ prev_in, float* __restrict__ prev_out,
                                                       const float* target, float* tout, const float* __restrict__ tprev_in, float* __restrict__ tprev_out,
                                                       const MixParams* __restrict__ p, int N, int CHW4, int W4, int H, int C4) {
  const int mode = p->mode;
  const float lam = p->lam, lam_real = p->lam_real;
  const int y1 = p->y1, y2 = p->y2, x1 = p->x1, x2 = p->x2;
  const size_t n_img = (size_t)N * CHW4, total = n_img + (size_t)N * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n_img) {
      const int n = (int)(i / CHW4), r = (int)(i - (size_t)n * CHW4);
      const f32x4 d = reinterpret_cast<const f32x4*>(data)[i];
      f32x4 o = d;
      if (mode != 0) {
        const f32x4 q = reinterpret_cast<const f32x4*>(prev_in)[(size_t)p->perm[n] * CHW4 + r];
        if (mode == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = lam * d[e] + (1.0f - lam) * q[e];
        } else {
          const int row = r / W4, x = (r - row * W4) * 4, y = row % H;
          const bool iny = y >= y1 && y < y2;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (iny && x + e >= x1 && x + e < x2) ? q[e] : d[e];
        }
      }
      if (mode != 0 || out != data) reinterpret_cast<f32x4*>(out)[i] = o;
      reinterpret_cast<f32x4*>(prev_out)[i] = d;
    } else {
      const size_t j = i - n_img;
      const int n = (int)(j / C4), r = (int)(j - (size_t)n * C4);
      const f32x4 t = reinterpret_cast<const f32x4*>(target)[j];
      f32x4 o = t;
      if (mode != 0) {
        const f32x4 q = reinterpret_cast<const f32x4*>(tprev_in)[(size_t)p->perm[n] * C4 + r];
        const float wc = mode == 1 ? lam : 1.0f - lam_real;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = wc * t[e] + (1.0f - wc) * q[e];
      }
      if (mode != 0 || tout != target) reinterpret_cast<f32x4*>(tout)[j] = o;
      reinterpret_cast<f32x4*>(tprev_out)[j] = t;
    }
  }
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" {

size_t mi355_mix_params_bytes(int N) { return sizeof(MixParams) + (size_t)(N > 0 ? N - 1 : 0) * sizeof(int); }

int mi355_mix_sample(void* params, unsigned long long seed, unsigned long long counter, int N, int H, int W, float cutmix_alpha,
                     float mixup_alpha, float prob, int allow, void* stream) {
  MI355_ARG(params && N >= 1 && N <= 1024 && H >= 1 && W >= 1, "mix_sample: N=%d (1..1024) H=%d W=%d", N, H, W);
  MI355_ARG(cutmix_alpha > 0 && mixup_alpha > 0 && allow >= 0 && allow <= 3, "mix_sample: alpha must be > 0, allow in 0..3");
  hipLaunchKernelGGL(mix_sample_kernel, dim3(1), dim3(N <= 64 ? 64 : 1024), 0, (hipStream_t)stream, reinterpret_cast<MixParams*>(params),
                     seed, counter, N, H, W, cutmix_alpha, mixup_alpha, prob, allow);
  MI355_LAUNCH_CHECK();
  return 0;
}

int mi355_mix_apply(const float* data, float* out, const float* prev_in, float* prev_out, const float* target, float* tout,
                    const float* tprev_in, float* tprev_out, const void* params, int N, int C, int H, int W, int num_classes, void* stream) {
  MI355_ARG(data && out && prev_in && prev_out && target && tout && tprev_in && tprev_out && params, "mix_apply: null pointer");
  MI355_ARG(prev_in != prev_out && tprev_in != tprev_out, "mix_apply: the previous-batch buffers must be double-buffered");
  MI355_ARG(N >= 1 && C >= 1 && H >= 1 && W % 4 == 0 && num_classes % 4 == 0, "mix_apply: W=%d and classes=%d must be multiples of 4", W,
            num_classes);
  const size_t total = (size_t)N * C * H * (W / 4) + (size_t)N * (num_classes / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(mix_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, data, out, prev_in, prev_out, target, tout, tprev_in, tprev_out,
                     reinterpret_cast<const MixParams*>(params), N, C * H * (W / 4), W / 4, H, num_classes / 4);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
