// bn.hip — BatchNorm2d (training + inference) forward/backward with fused ReLU / residual add, NHWC, gfx950.
//
// Replaces cuDNN BatchNormalization{ForwardTraining,Backward} + ATen relu/add/threshold_backward under
// `model(data)` / `loss.backward()` in the reference (sota_imagenet/callbacks.py:316-317); BN momentum
// semantics follow train.py:76 + arg_parser.py:132 (torch: running = (1-m)*running + m*batch, unbiased
// variance into running_var, eps inside the sqrt).
//
// All kernels are HBM-bound streams: each thread owns one 16-byte channel group (4 fp32 / 8 bf16 channels)
// and walks pixels, so every wave-instruction moves 1 KiB of contiguous NHWC memory.  Per-channel sums:
// registers -> wavefront __shfl_xor across the pixel rows that share a wave -> LDS across waves -> one
// partial row per block -> a finalize kernel that adds the (<= 512) block partials in fp64 in block order
// (bitwise reproducible; no float atomics).  In the training step most partial rows come from the conv kernels'
// epilogues instead (conv_igemm.hip, STATS 1 / 2) and only the finalize + apply kernels of this file run; the ReLU mask
// travels as one byte per 16-byte vector (written by bn_apply, read by both backward kernels); mask / residual /
// second-branch variants are template parameters and last-use streams are loaded non-temporally.
#include <hip/hip_fp8.h>

#include <algorithm>

#include "common.h"
#include "vec.h"

namespace mi355 {
namespace {

constexpr int MAXBLK = 512;

// fp8 training step: the e4m3 twin of the 8 bf16 values a thread has just computed (q = e4m3(bf16(v) * scale), saturating) and
// their running max magnitude; the twin is what the next fp8 convolution reads, the max seeds the next step's scale
__device__ __forceinline__ void quant8(const float (&v)[8], float scale, uint8_t* dst, float& amax) {
  // v_cvt_pk_fp8_f32 (OCP e4m3 on gfx950, round to nearest even) packs two values per instruction straight into the output
  // words; the clamp makes it saturating (inputs are finite) — no byte arrays, nothing goes through scratch
  float c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float r = (float)(bf16_t)v[e];
    amax = fmaxf(amax, fabsf(r));
    c[e] = __builtin_amdgcn_fmed3f(r * scale, 448.f, -448.f);
  }
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)lo, (unsigned)hi);
}
__device__ __forceinline__ void amax_flush(float amax, unsigned* dst) {
  // wave max -> workgroup max through LDS -> ONE atomic per workgroup, and only when it would raise the word (thousands of
  // same-address atomics serialise in one L2 channel: +85 us per launch when every wave issued its own)
  __shared__ float wmax[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > __uint_as_float(__atomic_load_n(dst, __ATOMIC_RELAXED))) atomicMax(dst, __float_as_uint(m));  // non-negative floats order like their bits
  }
}

// V (4 or 8) consecutive per-channel constants with 16-byte loads (c0 is a multiple of V): one or two load
// instructions instead of V — the per-thread prologue matters for the small late-layer tensors
template <int V>
__device__ __forceinline__ void load_consts(const float* __restrict__ p, int c0, float (&v)[V]) {
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + c0 + 4 * q);
    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
  }
}

// gradient wrt an ECA module's input from the gradient wrt the gated, drop-connect-scaled tensor: g * keep[n] * gate[n][c] + dpool[n][c],
// rounded to T as the stand-alone pass (variant.hip eca_residual_bwd_apply) stored it
template <typename T, int V>
__device__ __forceinline__ void eca_grad(const EcaGrad& eg, int n, int C, int c0, float (&gv)[V]) {
  const float kn = eg.keep ? eg.keep[n] : 1.f;
  const size_t o = (size_t)n * C + c0;
#pragma unroll
  for (int e = 0; e < V; ++e) gv[e] = (float)(T)(gv[e] * kn * eg.gate[o + e] + eg.dpool[o + e]);
}

struct ReduceArgs {
  const void* x;      // pre-BN tensor [M][C]
  const void* g;      // upstream gradient (bwd) or null
  const void* mask;   // post-activation tensor for the ReLU mask, or null
  const uint8_t* bits;  // or: the ReLU mask as one byte per 16-byte vector (bit e = element e was > 0), from bn_apply
  void* dz_out;       // optional masked gradient output
  const float* mean;  // [C] (bwd)
  const float* invstd;
  float* partial;     // [gridDim.x][2][C]
  float* pivot;       // [C] (stats): per-channel shift = the channel's value in row 0
  int M, C;
  float slope;        // MASK == 1: gradient factor where the activation was <= 0 (0 ReLU, 0.01 leaky ReLU)
  EcaGrad eg;         // MASK == 4: g stands for g * keep[n] * gate[n][c] + dpool[n][c] (the ECA backward applied on the fly)
};

// MODE 0: s1 = sum x, s2 = sum x^2.   MODE 1: s1 = sum dz, s2 = sum dz*xhat.
// MASK (MODE 1): 0 none, 1 post-activation tensor (slope), 2 bit mask (ReLU), 3 bit mask + slope (leaky ReLU);  DZ: also store the masked gradient.
// The variants are template parameters, not runtime branches: a branch in the loop body keeps the compiler from
// hoisting the loads of the unrolled iterations above the arithmetic, and these kernels live on loads in flight.
template <typename T, int MODE, int MASK, bool DZ>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const ReduceArgs p) {
  constexpr int V = Vec16<T>::N;
  __shared__ __attribute__((aligned(16))) float red[2][256 * V];
  const int tpr_full = p.C / V;
  const int tpr = tpr_full < 256 ? tpr_full : 256;
  const int rpp = 256 / tpr;
  const int tid = threadIdx.x;
  const int cv = tid % tpr, r = tid / tpr;
  const int c0 = (blockIdx.y * tpr + cv) * V;
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* g = reinterpret_cast<const T*>(p.g);
  const T* mk = reinterpret_cast<const T*>(p.mask);
  T* dz_out = reinterpret_cast<T*>(p.dz_out);

  float s1[V], s2[V], mu[V], is[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    s1[e] = 0.f;
    s2[e] = 0.f;
    mu[e] = 0.f;
    is[e] = 0.f;
  }
  if (MODE == 1) {
    load_consts<V>(p.mean, c0, mu);
    load_consts<V>(p.invstd, c0, is);
  } else {
    // shifted sums: var = E[(x-p)^2] - E[x-p]^2 with p a sample of the channel => no catastrophic
    // cancellation when |mean| >> std (the oracle's torch-CPU path accumulates in fp64)
    Vec16<T>::load(x + c0, mu);
    if (blockIdx.x == 0 && r == 0) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        p.pivot[c0 + e] = mu[e];
      }
    }
  }
  const int step = gridDim.x * rpp;
#pragma unroll 4
  for (int m = blockIdx.x * rpp + r; m < p.M; m += step) {
    const size_t off = (size_t)m * p.C + c0;
    float xv[V];
    Vec16<T>::load(x + off, xv);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float d = xv[e] - mu[e];
        s1[e] += d;
        s2[e] += d * d;
      }
    } else {
      float gv[V];
      Vec16<T>::load(g + off, gv);
      if constexpr (MASK == 2) {
        const unsigned b = p.bits[off / V];
#pragma unroll
        for (int e = 0; e < V; ++e) gv[e] = (b >> e) & 1u ? gv[e] : 0.f;
      } else if constexpr (MASK == 3) {  // bit mask of a leaky ReLU
        const unsigned b = p.bits[off / V];
#pragma unroll
        for (int e = 0; e < V; ++e) gv[e] = (b >> e) & 1u ? gv[e] : gv[e] * p.slope;
      } else if constexpr (MASK == 1) {
        float mv[V];
        Vec16<T>::load(mk + off, mv);
#pragma unroll
        for (int e = 0; e < V; ++e) gv[e] = mv[e] > 0.f ? gv[e] : gv[e] * p.slope;  // slope 0: ReLU, 0.01: leaky ReLU
      } else if constexpr (MASK == 4) {
        eca_grad<T, V>(p.eg, m / p.eg.hw, p.C, c0, gv);
      }
      if constexpr (DZ) Vec16<T>::store(dz_out + off, gv);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (xv[e] - mu[e]) * is[e];
        s1[e] += gv[e];
        s2[e] += gv[e] * xh;
      }
    }
  }
  // rows that share a wavefront: shuffle-reduce, then one LDS row per wave; otherwise one LDS row per r
  int nrows, myrow;
  bool writer;
  if (tpr < 64) {
    for (int off = tpr; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        s1[e] += __shfl_xor(s1[e], off);
        s2[e] += __shfl_xor(s2[e], off);
      }
    }
    nrows = 4;
    myrow = tid >> 6;
    writer = (tid & 63) < tpr;
  } else {
    nrows = rpp;
    myrow = r;
    writer = true;
  }
  if (writer) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red[0][(myrow * tpr + cv) * V + e] = s1[e];
      red[1][(myrow * tpr + cv) * V + e] = s2[e];
    }
  }
  __syncthreads();
  if (tid < tpr) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float a = 0.f, b = 0.f;
      for (int rr = 0; rr < nrows; ++rr) {
        a += red[0][(rr * tpr + tid) * V + e];
        b += red[1][(rr * tpr + tid) * V + e];
      }
      const int c = (blockIdx.y * tpr + tid) * V + e;
      p.partial[((size_t)blockIdx.x * 2 + 0) * p.C + c] = a;
      p.partial[((size_t)blockIdx.x * 2 + 1) * p.C + c] = b;
    }
  }
}

// 16 channels x 16 slices of the block partials per workgroup; fp64 sums in fixed order.
struct FinalizeArgs {
  const float* partial;
  const float* pivot;
  int nblk, M, C;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float* scale;
  float* shift;
  float eps, momentum;
  // bwd
  const float* invstd;
  float* dgamma;
  float* dbeta;
  float beta_acc;
  float* coef;
};

// rule 2 of the comment below (the A/B switch MI355_BN_FIN_WIDE is gone: its verdict is profiles/r05_bn_finalize_per_launch.txt and the two same-box
// step measurements quoted there)
static int fin_wide_mode() { return 2; }
// one channel per workgroup?  0 never; 1: few channels and more than one pass of rows (round 5's first rule); 2 (default): also up to 512
// channels from 128 rows on — four-channel workgroups leave most of the chip idle there and the kernel is a chain of load latencies
// (same box: 18.15 -> 18.05 ms per step, 17.75 -> 17.69 on a faster box); 3 / 4: wider still (no further gain)
constexpr int FIN_WIDE_ROWS = 4096;
static bool fin_one_channel(int C, int nblk) {
  const int m = fin_wide_mode();
  if (m <= 0) return false;
  if (m == 1) return C <= 128 && nblk > 512;
  if (m == 3) return (C <= 128 && nblk > 512) || (C <= 1024 && nblk >= 64);
  if (m == 4) return true;
  return (C <= 128 && nblk > 512) || (C <= 512 && nblk >= 128);
}

template <int MODE, int CPW = 4, int BT = 256>
__global__ __launch_bounds__(BT) void bn_finalize_kernel(const FinalizeArgs p) {
  // CPW channels x 256 / CPW slices per workgroup: every thread has <= 8 independent partial rows to add per pass, then a fixed-shape
  // LDS tree — the summation order depends only on nblk (and CPW, itself a function of C and nblk), so results are bitwise reproducible.
  // CPW = 4: 64 slices, one pass up to 512 rows.  CPW = 1 (few channels, many rows: the 64-channel layers whose direct-conv kernels
  // leave one row per 4-row tile, 3584 at batch 256): 256 slices, so 3584 rows are two passes instead of seven dependent ones — the
  // kernel is a chain of load latencies, and beside an HBM-bound weight-gradient kernel each pass takes 10 us
  // (profiles/r05_bn_finalize_per_launch.txt: 35 us serial / 70-150 us in the step -> 8 / 15).
  // BT = 1024 (CPW = 1, more than FIN_WIDE_ROWS rows: the 112 x 112 x 64 layers of BASELINE configs[3] on two-row tiles leave 14 336): 1024 slices, two
  // passes instead of seven (59 -> ~20 us per launch).  The shape is a function of (C, nblk) like CPW: every path that finalizes a layer takes the same one.
  constexpr int SL = BT / CPW;
  __shared__ double red[2][SL][CPW];
  const int cl = threadIdx.x % CPW, sl = threadIdx.x / CPW;
  const int c = blockIdx.x * CPW + cl;
  double a = 0.0, b = 0.0;
  for (int k0 = sl; k0 < p.nblk; k0 += SL * 8) {
    float va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + SL * u;
      const bool in = k < p.nblk;
      va[u] = in ? p.partial[((size_t)k * 2 + 0) * p.C + c] : 0.f;
      vb[u] = in ? p.partial[((size_t)k * 2 + 1) * p.C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a += (double)va[u];
      b += (double)vb[u];
    }
  }
  red[0][sl][cl] = a;
  red[1][sl][cl] = b;
  __syncthreads();
#pragma unroll
  for (int s = SL / 2; s > 0; s >>= 1) {
    if (sl < s) {
      red[0][sl][cl] += red[0][sl + s][cl];
      red[1][sl][cl] += red[1][sl + s][cl];
    }
    __syncthreads();
  }
  if (sl == 0) {
    const double s1 = red[0][0][cl], s2 = red[1][0][cl];
    const double M = (double)p.M;
    if (MODE == 0) {
      const double dm = s1 / M;  // mean of (x - pivot)
      const double mean = (p.pivot ? (double)p.pivot[c] : 0.0) + dm;
      double var = s2 / M - dm * dm;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)p.eps));
      const float meanf = (float)mean;
      p.save_mean[c] = meanf;
      p.save_invstd[c] = invstd;
      const float sc = p.gamma[c] * invstd;
      p.scale[c] = sc;
      p.shift[c] = p.beta[c] - meanf * sc;
      if (p.running_mean) {
        const double unb = p.M > 1 ? var * M / (M - 1.0) : var;
        p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * meanf;
        p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unb;
      }
    } else {
      const float db = (float)s1, dg = (float)s2;
      p.dbeta[c] = (p.beta_acc != 0.f ? p.beta_acc * p.dbeta[c] : 0.f) + db;
      p.dgamma[c] = (p.beta_acc != 0.f ? p.beta_acc * p.dgamma[c] : 0.f) + dg;
      p.coef[c] = p.gamma[c] * p.invstd[c];
      p.coef[p.C + c] = (float)(s1 / M);
      p.coef[2 * p.C + c] = (float)(s2 / M);
    }
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float* scale, float* shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

struct ApplyArgs {
  const void* x;
  const float* scale;
  const float* shift;
  const void* residual;
  const void* x2;
  const float* scale2;
  const float* shift2;
  void* out;
  uint8_t* bits;  // optional ReLU mask output: one byte per 16-byte vector
  size_t nvec;    // M*C/V
  int cvecs;      // C/V
  int relu;
  float slope;    // of the negative side (RELU == 1 only): 0 ReLU, 0.01 leaky ReLU
  QuantOut qo;    // Q: e4m3 twin of `out`
};

// RES: + residual;  X2: + second normalised tensor (downsample branch);  RELU: 0 none, 1 (leaky) relu with p.slope, 2 relu + bit mask out,
// 3 leaky relu with p.slope + bit mask out (the BResNet-50 executor)
// Q (bf16 only): also the e4m3 twin of the output + its amax (fp8 training step)
template <typename T, bool RES, bool X2, int RELU, bool Q = false, bool QONLY = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const ApplyArgs p) {
  constexpr int V = Vec16<T>::N;
  static_assert(!Q || V == 8, "the quantised twin exists for bf16 tensors");
  float qs = 1.f, amax = 0.f;
  if constexpr (Q) qs = p.qo.scale[0];
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* res = reinterpret_cast<const T*>(p.residual);
  const T* x2 = reinterpret_cast<const T*>(p.x2);
  T* out = reinterpret_cast<T*>(p.out);
  const size_t stride = (size_t)gridDim.x * 256;  // multiple of cvecs (host guarantees)
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(i % p.cvecs) * V;
  float sc[V], sh[V], sc2[V], sh2[V];
  load_consts<V>(p.scale, c0, sc);
  load_consts<V>(p.shift, c0, sh);
  if constexpr (X2) {
    load_consts<V>(p.scale2, c0, sc2);
    load_consts<V>(p.shift2, c0, sh2);
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      sc2[e] = 0.f;
      sh2[e] = 0.f;
    }
  }
#pragma unroll 2
  for (; i < p.nvec; i += stride) {
    float v[V];
    Vec16<T>::load_nt(x + i * V, v);  // y is not read again before backward
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
    if constexpr (RES) {
      float rv[V];
      Vec16<T>::load_nt(res + i * V, rv);  // the block input: next read in backward
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] += rv[e];
    }
    if constexpr (X2) {
      float rv[V];
      Vec16<T>::load_nt(x2 + i * V, rv);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] += fmaf(rv[e], sc2[e], sh2[e]);
    }
    if constexpr (RELU != 0) {
      if constexpr (RELU == 2 || RELU == 3) {
        unsigned b = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
        __builtin_nontemporal_store((uint8_t)b, p.bits + i);  // read in backward only
      }
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = v[e] > 0.f ? v[e] : (RELU == 2 ? 0.f : v[e] * p.slope);
    }
    if constexpr (!QONLY) Vec16<T>::store(out + i * V, v);
    if constexpr (Q) quant8(v, qs, p.qo.q + i * V, amax);
  }
  if constexpr (Q) amax_flush(amax, p.qo.amax);
}

struct BwdApplyArgs {
  const void* g;
  const void* mask;
  const uint8_t* bits;
  const void* x;
  const float* mean;
  const float* invstd;
  const float* coef;  // [3][C]: gamma*invstd, mean(dz), mean(dz*xhat)
  void* dx;
  size_t nvec;
  int cvecs, C;
  float slope;  // MASK == 1: gradient factor where the activation was <= 0
  EcaGrad eg;   // MASK == 4 (see ReduceArgs)
  QuantOut qo;  // Q: e4m3 twin of dx
};

template <typename T, int MASK, bool Q = false, bool QONLY = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BwdApplyArgs p) {
  constexpr int V = Vec16<T>::N;
  static_assert(!Q || V == 8, "the quantised twin exists for bf16 tensors");
  float qs = 1.f, amax = 0.f;
  if constexpr (Q) qs = p.qo.scale[0];
  const T* g = reinterpret_cast<const T*>(p.g);
  const T* mk = reinterpret_cast<const T*>(p.mask);
  const T* x = reinterpret_cast<const T*>(p.x);
  T* dx = reinterpret_cast<T*>(p.dx);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(i % p.cvecs) * V;
  float mu[V], is[V], k0[V], k1[V], k2[V];
  load_consts<V>(p.mean, c0, mu);
  load_consts<V>(p.invstd, c0, is);
  load_consts<V>(p.coef, c0, k0);
  load_consts<V>(p.coef + p.C, c0, k1);
  load_consts<V>(p.coef + 2 * p.C, c0, k2);
#pragma unroll 2
  for (; i < p.nvec; i += stride) {
    float gv[V], xv[V];
    Vec16<T>::load_nt(g + i * V, gv);  // last use of g and of y
    Vec16<T>::load_nt(x + i * V, xv);
    if constexpr (MASK == 2) {
      const unsigned b = p.bits[i];
#pragma unroll
      for (int e = 0; e < V; ++e) gv[e] = (b >> e) & 1u ? gv[e] : 0.f;
    } else if constexpr (MASK == 3) {
      const unsigned b = p.bits[i];
#pragma unroll
      for (int e = 0; e < V; ++e) gv[e] = (b >> e) & 1u ? gv[e] : gv[e] * p.slope;
    } else if constexpr (MASK == 4) {
      eca_grad<T, V>(p.eg, (int)((i / p.cvecs) / (size_t)p.eg.hw), p.C, c0, gv);
    } else if constexpr (MASK == 1) {
      float mv[V];
      Vec16<T>::load(mk + i * V, mv);
#pragma unroll
      for (int e = 0; e < V; ++e) gv[e] = mv[e] > 0.f ? gv[e] : gv[e] * p.slope;
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float xh = (xv[e] - mu[e]) * is[e];
      gv[e] = k0[e] * (gv[e] - k1[e] - xh * k2[e]);
    }
    if constexpr (!QONLY) Vec16<T>::store(dx + i * V, gv);
    if constexpr (Q) quant8(gv, qs, p.qo.q + i * V, amax);
  }
  if constexpr (Q) amax_flush(amax, p.qo.amax);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Stem backward without the full-resolution gradient of the max pool: g = maxpool3x3/2-backward(dp) is GATHERED on the fly from
// the pooled gradient dp [N][H/2][W/2][64] and the argmax codes (1 byte per pooled element, position kh * 3 + kw inside the
// window) by both BN-backward passes, instead of being written by a pool-backward kernel and read back twice (2 x 411 MB at
// batch 256 / 224 px; maxpool_bwd_kernel in misc.hip is the stand-alone operator, same arithmetic).  One thread per 2 x 2
// block of full-resolution pixels and channel vector: the windows (a, b) ... (a + 1, b + 1) are the only ones that reach the
// block, a pixel adds its windows in (oh, ow) order and the sum is rounded to T — the value the unfused path stored.
// Register diet (these loops live on loads in flight; the first version needed 227 VGPRs = two waves per SIMD and ran at
// half the speed of the apply pass): every stream goes through a raw buffer resource with ONE 32-bit offset register per tensor
// (row / window steps in the scalar offset and the immediate), gradients and codes stay packed until the pixel that uses them.
constexpr int SC = 64;  // channels of the stage (the stem)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xfffffffcull ? 0xfffffffcull : bytes), 0x00020000);
}
__device__ __forceinline__ uint4 buf16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool nt = false) {
  return nt ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2))   // slc
            : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <typename T>
struct PoolGather {
  static constexpr int V = Vec16<T>::N;
  uint4 g[4];
  uint32_t id[4][V / 4];
  // window w = 2 dr + dc sits (dr * Wo + dc) pooled pixels behind (a, b); a window off the edge is read anyway (the next row /
  // image: mapped memory; behind the tensor: the buffer range check) and its codes are voided
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rg, __amdgpu_buffer_rsrc_t ri, unsigned t, unsigned cv, int a, int b, int Ho, int Wo) {
    const unsigned go = (t * SC + cv * V) * (unsigned)sizeof(T), io = t * SC + cv * V;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const bool ok = a + (w >> 1) < Ho && b + (w & 1) < Wo;
      const unsigned srow = (w >> 1) ? (unsigned)Wo * SC : 0u;
      g[w] = buf16(rg, go + (w & 1) * SC * (unsigned)sizeof(T), srow * (unsigned)sizeof(T));
      if constexpr (V == 4) {
        const unsigned x = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ri, io + (w & 1) * SC, srow, 0);
        id[w][0] = ok ? x : 0xffffffffu;  // (no position has code 255)
      } else {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 x = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(ri, io + (w & 1) * SC, srow, 0));
        id[w][0] = ok ? x[0] : 0xffffffffu;
        id[w][1] = ok ? x[1] : 0xffffffffu;
      }
    }
  }
  // gradient of block pixel (r, c), under its ReLU bits
  template <int r, int c>
  __device__ __forceinline__ void pixel(unsigned bits, float (&out)[V]) const {
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = 0.f;
#pragma unroll
    for (int dr = 0; dr <= r; ++dr)
#pragma unroll
      for (int dc = 0; dc <= c; ++dc) {
        const unsigned pos = (r + 1 - 2 * dr) * 3 + (c + 1 - 2 * dc);
        float gw[V];
        Vec16<T>::unpack(g[2 * dr + dc], gw);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (((id[2 * dr + dc][e >> 2] >> (8 * (e & 3))) & 0xffu) == pos) out[e] += gw[e];
      }
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = (bits >> e) & 1u ? (float)(T)out[e] : 0.f;
  }
};

struct StemBwdArgs {
  const void* dp;        // [N][Ho][Wo][64] gradient wrt the pooled activation
  const uint8_t* idx;    // [N][Ho][Wo][64] argmax codes
  const uint8_t* bits;   // [N][2Ho][2Wo][64 / V] ReLU bits of the full-resolution activation
  const void* y;         // [N][2Ho][2Wo][64] raw conv output
  const float* mean;
  const float* invstd;
  const float* coef;     // apply: [3][64]
  float* partial;        // reduce: [gridDim.x][2][64]
  void* dx;              // apply: gradient wrt y
  int N, Ho, Wo;
};

// the four pixels of block t = (n, a, b): y vectors (last use in the apply pass: streamed) and ReLU bits
template <typename T>
struct BlockIn {
  uint4 y[4];
  unsigned bt[4];
  unsigned pix00;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t ry, __amdgpu_buffer_rsrc_t rb, int n, int a, int b, int Ho, int Wo, unsigned cv, bool nt) {
    constexpr int V = Vec16<T>::N;
    const unsigned W = 2 * Wo;
    pix00 = ((unsigned)n * 2 * Ho + 2 * a) * W + 2 * b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned srow = (k >> 1) ? W : 0u;
      y[k] = buf16(ry, (pix00 * SC + cv * V) * (unsigned)sizeof(T) + (k & 1) * SC * (unsigned)sizeof(T), srow * SC * (unsigned)sizeof(T), nt);
      bt[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rb, pix00 * (SC / V) + cv + (k & 1) * (SC / V), srow * (SC / V), 0);
    }
  }
};

// pass 1: s1 = sum dz, s2 = sum dz * xhat.  256 threads = 64 / V vectors x 256 / (64 / V) block slots
template <typename T>
__global__ __launch_bounds__(256, 2) void stem_bwd_reduce_kernel(const StemBwdArgs p) {
  constexpr int V = Vec16<T>::N, tpr = SC / V, rpp = 256 / tpr;
  __shared__ __attribute__((aligned(16))) float red[2][4 * SC];
  const int tid = threadIdx.x;
  const unsigned cv = tid % tpr;
  const int r = tid / tpr;
  const size_t full = (size_t)p.N * p.Ho * p.Wo * 4 * SC;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(p.y, full * sizeof(T)), rb = rsrc_of(p.bits, full / V);
  const __amdgpu_buffer_rsrc_t rg = rsrc_of(p.dp, full / 4 * sizeof(T)), ri = rsrc_of(p.idx, full / 4);
  float s1[V], s2[V], mu[V], is[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s1[e] = s2[e] = 0.f;
  load_consts<V>(p.mean, cv * V, mu);
  load_consts<V>(p.invstd, cv * V, is);
  const int nb = p.N * p.Ho * p.Wo;
  const int step = gridDim.x * rpp;
  // the NEXT block's 16 loads are issued before this block's ~600 instructions of arithmetic: a wave then always has a batch in
  // flight (three waves per SIMD alone leave HBM idle while they compute)
  auto fetch = [&](PoolGather<T>& pg, BlockIn<T>& in, int t) __attribute__((always_inline)) {
    const int b = t % p.Wo, a = (t / p.Wo) % p.Ho, n = t / (p.Wo * p.Ho);
    pg.load(rg, ri, (unsigned)t, cv, a, b, p.Ho, p.Wo);
    in.load(ry, rb, n, a, b, p.Ho, p.Wo, cv, false);
  };
  auto consume = [&](const PoolGather<T>& pg, const BlockIn<T>& in) __attribute__((always_inline)) {
    auto add = [&](const float (&gv)[V], const uint4& yraw) __attribute__((always_inline)) {
      float yv[V];
      Vec16<T>::unpack(yraw, yv);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (yv[e] - mu[e]) * is[e];
        s1[e] += gv[e];
        s2[e] += gv[e] * xh;
      }
    };
    float gv[V];
    pg.template pixel<0, 0>(in.bt[0], gv);
    add(gv, in.y[0]);
    pg.template pixel<0, 1>(in.bt[1], gv);
    add(gv, in.y[1]);
    pg.template pixel<1, 0>(in.bt[2], gv);
    add(gv, in.y[2]);
    pg.template pixel<1, 1>(in.bt[3], gv);
    add(gv, in.y[3]);
  };
  int t = blockIdx.x * rpp + r;
  if (t < nb) {
    PoolGather<T> pga, pgb;
    BlockIn<T> ina, inb;
    fetch(pga, ina, t);
#pragma unroll 1
    for (;;) {
      int tn = t + step;
      if (tn < nb) fetch(pgb, inb, tn);
      consume(pga, ina);
      if (tn >= nb) break;
      t = tn + step;
      if (t < nb) fetch(pga, ina, t);
      consume(pgb, inb);
      if (t >= nb) break;
    }
  }
  // block slots that share a wavefront: shuffle-reduce, then one LDS row per wave
  for (int off = tpr; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      s1[e] += __shfl_xor(s1[e], off);
      s2[e] += __shfl_xor(s2[e], off);
    }
  }
  if ((tid & 63) < tpr) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red[0][(tid >> 6) * SC + cv * V + e] = s1[e];
      red[1][(tid >> 6) * SC + cv * V + e] = s2[e];
    }
  }
  __syncthreads();
  if (tid < SC) {
    float a = 0.f, b = 0.f;
    for (int rr = 0; rr < 4; ++rr) {  // the waves, in order
      a += red[0][rr * SC + tid];
      b += red[1][rr * SC + tid];
    }
    p.partial[((size_t)blockIdx.x * 2 + 0) * SC + tid] = a;
    p.partial[((size_t)blockIdx.x * 2 + 1) * SC + tid] = b;
  }
}

// pass 2: dx = k0 * (dz - k1 - xhat * k2)
template <typename T>
__global__ __launch_bounds__(256, 4) void stem_bwd_apply_kernel(const StemBwdArgs p) {
  constexpr int V = Vec16<T>::N, tpr = SC / V;
  T* dx = reinterpret_cast<T*>(p.dx);
  const size_t full = (size_t)p.N * p.Ho * p.Wo * 4 * SC;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(p.y, full * sizeof(T)), rb = rsrc_of(p.bits, full / V);
  const __amdgpu_buffer_rsrc_t rg = rsrc_of(p.dp, full / 4 * sizeof(T)), ri = rsrc_of(p.idx, full / 4);
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  const unsigned stride = gridDim.x * 256;  // a multiple of tpr: a thread keeps its channel vector
  const unsigned cv = i % tpr, c0 = cv * V;
  float mu[V], is[V], k0[V], k1[V], k2[V];
  load_consts<V>(p.mean, c0, mu);
  load_consts<V>(p.invstd, c0, is);
  load_consts<V>(p.coef, c0, k0);
  load_consts<V>(p.coef + SC, c0, k1);
  load_consts<V>(p.coef + 2 * SC, c0, k2);
  const unsigned total = (unsigned)p.N * p.Ho * p.Wo * tpr;
  const unsigned W = 2 * p.Wo;
#pragma unroll 1
  for (; i < total; i += stride) {
    const int t = (int)(i / tpr);
    const int b = t % p.Wo, a = (t / p.Wo) % p.Ho, n = t / (p.Wo * p.Ho);
    PoolGather<T> pg;
    pg.load(rg, ri, (unsigned)t, cv, a, b, p.Ho, p.Wo);
    BlockIn<T> in;
    in.load(ry, rb, n, a, b, p.Ho, p.Wo, cv, true);
    auto put = [&](float (&gv)[V], const uint4& yraw, int k) __attribute__((always_inline)) {
      float yv[V];
      Vec16<T>::unpack(yraw, yv);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (yv[e] - mu[e]) * is[e];
        gv[e] = k0[e] * (gv[e] - k1[e] - xh * k2[e]);
      }
      Vec16<T>::store(dx + ((size_t)in.pix00 + (k >> 1) * W + (k & 1)) * SC + c0, gv);
    };
    float gv[V];
    pg.template pixel<0, 0>(in.bt[0], gv);
    put(gv, in.y[0], 0);
    pg.template pixel<0, 1>(in.bt[1], gv);
    put(gv, in.y[1], 1);
    pg.template pixel<1, 0>(in.bt[2], gv);
    put(gv, in.y[2], 2);
    pg.template pixel<1, 1>(in.bt[3], gv);
    put(gv, in.y[3], 3);
  }
}

int reduce_grid(int dtype, int M, int C, dim3* grid) {
  const int V = 16 / (int)dtype_size(dtype);
  const int tpr_full = C / V;
  const int tpr = tpr_full < 256 ? tpr_full : 256;
  const int rpp = 256 / tpr;
  const int gy = tpr_full / tpr;
  int nblk = cdiv(M, rpp * 8);
  // partial rows of the stand-alone reductions: MAXBLK (two workgroups per CU; 1024 rows measured +0.2 ms per step in round 1)
  const int cap = MAXBLK;
  if (nblk > cap / gy) nblk = cap / gy;
  if (nblk < 1) nblk = 1;
  *grid = dim3(nblk, gy);
  return nblk;
}

int elementwise_blocks(size_t nvec, int cvecs) {
  // 4 vectors per thread (amortises the per-channel constant loads; 2 is a tie, 8 and 1 are slower) and NO practical cap on the workgroups: the cap of
  // 4096 the kernels ran under until late round 5 cost 0.15 ms per bf16 step and 0.47 ms per fp8 step at batch 512 (17.72 -> 17.55, 33.8 -> 33.3, same box:
  // short workgroups interleave with the weight-gradient stream's kernels and leave no tail; profiles/r05b_ab_elementwise_grid_caps.txt).
  const size_t vpt = 4;
  size_t b = (nvec + vpt * 256 - 1) / (vpt * 256);
  const size_t cap = 65536;
  if (b > cap) b = cap;
  if (b < 4) b = 4;
  // stride = b*256 must be a multiple of cvecs (a power of two <= 1024): make b a multiple of 4
  b = (b + 3) / 4 * 4;
  (void)cvecs;
  return (int)b;
}

int check_c(int dtype, int C) {
  MI355_ARG(dtype == MI355_F32 || dtype == MI355_BF16, "bn: bad dtype %d", dtype);
  MI355_ARG(C >= 64 && (C & (C - 1)) == 0 && C <= 4096, "bn: C=%d must be a power of two in [64, 4096]", C);
  return 0;
}

}  // namespace

int bn_max_blocks() { return 2 * MAXBLK; }  // conv-epilogue statistics use 2 partial rows per workgroup (<= 1024)

int launch_bn_stats(int dtype, const void* x, float* partial, float* pivot, int* nblk_out, int M, int C,
                    hipStream_t s) {
  MI355_TRY(check_c(dtype, C));
  ReduceArgs a{};
  a.x = x;
  a.partial = partial;
  a.pivot = pivot;
  a.M = M;
  a.C = C;
  dim3 grid;
  *nblk_out = reduce_grid(dtype, M, C, &grid);
  if (dtype == MI355_F32)
    hipLaunchKernelGGL((bn_reduce_kernel<float, 0, 0, false>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 0, 0, false>), grid, dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_finalize(const float* partial, const float* pivot, int nblk, int M, int C, const float* gamma,
                       const float* beta,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                       float* scale, float* shift, float eps, float momentum, hipStream_t s) {
  FinalizeArgs a{};
  a.partial = partial;
  a.pivot = pivot;
  a.nblk = nblk;
  a.M = M;
  a.C = C;
  a.gamma = gamma;
  a.beta = beta;
  a.running_mean = running_mean;
  a.running_var = running_var;
  a.save_mean = save_mean;
  a.save_invstd = save_invstd;
  a.scale = scale;
  a.shift = shift;
  a.eps = eps;
  a.momentum = momentum;
  if (fin_one_channel(C, nblk) && nblk > FIN_WIDE_ROWS) hipLaunchKernelGGL((bn_finalize_kernel<0, 1, 1024>), dim3(C), dim3(1024), 0, s, a);
  else if (fin_one_channel(C, nblk)) hipLaunchKernelGGL((bn_finalize_kernel<0, 1>), dim3(C), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((bn_finalize_kernel<0, 4>), dim3(C / 4), dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float* scale,
                          float* shift, int C, float eps, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, rm, rv, scale, shift, C,
                     eps);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                    const void* x2, const float* scale2, const float* shift2, void* out, int M, int C, int relu,
                    hipStream_t s, uint8_t* relu_bits, QuantOut qo) {
  MI355_TRY(check_c(dtype, C));
  const int V = 16 / (int)dtype_size(dtype);
  ApplyArgs a{};
  a.x = x;
  a.scale = scale;
  a.shift = shift;
  a.residual = residual;
  a.x2 = x2;
  a.scale2 = scale2;
  a.shift2 = shift2;
  a.out = out;
  a.bits = relu_bits;
  a.nvec = (size_t)M * C / V;
  a.cvecs = C / V;
  a.relu = relu;
  a.slope = relu == 2 ? 0.01f : 0.f;  // activation code 2 = leaky ReLU (per-op path; the bit-mask path is ReLU)
  a.qo = qo;
  const int blocks = elementwise_blocks(a.nvec, a.cvecs);
  if (qo.q) {  // fp8 training step: bf16, ReLU with bit mask (the executor's training forward)
    MI355_ARG(dtype == MI355_BF16 && relu == 1 && relu_bits && qo.scale && qo.amax && !(residual && x2), "bn_apply: the e4m3 twin needs bf16 + ReLU bit mask");
    if (residual) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, true, false, 2, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (x2) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, true, 2, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (qo.only) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, false, 2, true, true>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, false, 2, true>), dim3(blocks), dim3(256), 0, s, a);
    MI355_LAUNCH_CHECK();
    return 0;
  }
  const int rl = relu ? (relu_bits ? 2 : 1) : 0;
  const int variant = (residual ? 6 : 0) + (x2 ? 3 : 0) + rl;
  MI355_ARG(!(residual && x2), "bn_apply: residual and second branch together are not supported");
  if (relu == 2 && relu_bits) {  // leaky ReLU + bit mask
    MI355_ARG(!residual && !x2, "bn_apply: the leaky bit-mask form has no residual / second-branch variant");
    if (dtype == MI355_F32) hipLaunchKernelGGL((bn_apply_kernel<float, false, false, 3>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, false, 3>), dim3(blocks), dim3(256), 0, s, a);
    MI355_LAUNCH_CHECK();
    return 0;
  }
#define MI355_BN_APPLY(TT)                                                                                      \
  switch (variant) {                                                                                            \
    case 0: hipLaunchKernelGGL((bn_apply_kernel<TT, false, false, 0>), dim3(blocks), dim3(256), 0, s, a); break; \
    case 1: hipLaunchKernelGGL((bn_apply_kernel<TT, false, false, 1>), dim3(blocks), dim3(256), 0, s, a); break; \
    case 2: hipLaunchKernelGGL((bn_apply_kernel<TT, false, false, 2>), dim3(blocks), dim3(256), 0, s, a); break; \
    case 3: hipLaunchKernelGGL((bn_apply_kernel<TT, false, true, 0>), dim3(blocks), dim3(256), 0, s, a); break;  \
    case 4: hipLaunchKernelGGL((bn_apply_kernel<TT, false, true, 1>), dim3(blocks), dim3(256), 0, s, a); break;  \
    case 5: hipLaunchKernelGGL((bn_apply_kernel<TT, false, true, 2>), dim3(blocks), dim3(256), 0, s, a); break;  \
    case 6: hipLaunchKernelGGL((bn_apply_kernel<TT, true, false, 0>), dim3(blocks), dim3(256), 0, s, a); break;  \
    case 7: hipLaunchKernelGGL((bn_apply_kernel<TT, true, false, 1>), dim3(blocks), dim3(256), 0, s, a); break;  \
    default: hipLaunchKernelGGL((bn_apply_kernel<TT, true, false, 2>), dim3(blocks), dim3(256), 0, s, a); break; \
  }
  if (dtype == MI355_F32) {
    MI355_BN_APPLY(float)
  } else {
    MI355_BN_APPLY(bf16_t)
  }
#undef MI355_BN_APPLY
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_bwd_reduce(int dtype, const void* g, const void* mask_src, const void* x, const float* mean,
                         const float* invstd, void* dz_out, float* partial, int* nblk_out, int M, int C,
                         hipStream_t s, const uint8_t* relu_bits, float slope, const EcaGrad* eg) {
  MI355_TRY(check_c(dtype, C));
  ReduceArgs a{};
  if (eg) a.eg = *eg;
  a.slope = slope;
  a.x = x;
  a.g = g;
  a.mask = mask_src;
  a.bits = relu_bits;
  a.dz_out = dz_out;
  a.mean = mean;
  a.invstd = invstd;
  a.partial = partial;
  a.M = M;
  a.C = C;
  dim3 grid;
  *nblk_out = reduce_grid(dtype, M, C, &grid);
  const int mask = relu_bits ? (slope != 0.f ? 3 : 2) : mask_src ? 1 : 0;
  if (eg) {  // the gradient comes out of an ECA backward on the fly: identity activation, nothing else stored
    MI355_ARG(mask == 0 && !dz_out && eg->gate && eg->dpool && eg->hw > 0, "bn_bwd_reduce: the ECA gradient source takes no mask / dz output");
    if (dtype == MI355_F32) hipLaunchKernelGGL((bn_reduce_kernel<float, 1, 4, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 1, 4, false>), grid, dim3(256), 0, s, a);
    MI355_LAUNCH_CHECK();
    return 0;
  }
  const int variant = mask * 2 + (dz_out ? 1 : 0);
#define MI355_BN_REDUCE(TT)                                                                                     \
  switch (variant) {                                                                                            \
    case 0: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 0, false>), grid, dim3(256), 0, s, a); break;           \
    case 1: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 0, true>), grid, dim3(256), 0, s, a); break;            \
    case 2: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 1, false>), grid, dim3(256), 0, s, a); break;           \
    case 3: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 1, true>), grid, dim3(256), 0, s, a); break;            \
    case 4: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 2, false>), grid, dim3(256), 0, s, a); break;           \
    case 5: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 2, true>), grid, dim3(256), 0, s, a); break;            \
    case 6: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 3, false>), grid, dim3(256), 0, s, a); break;           \
    default: hipLaunchKernelGGL((bn_reduce_kernel<TT, 1, 3, true>), grid, dim3(256), 0, s, a); break;           \
  }
  if (dtype == MI355_F32) {
    MI355_BN_REDUCE(float)
  } else {
    MI355_BN_REDUCE(bf16_t)
  }
#undef MI355_BN_REDUCE
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_bwd_finalize(const float* partial, int nblk, int M, int C, const float* gamma, const float* invstd,
                           float* dgamma, float* dbeta, float beta_acc, float* coef, hipStream_t s) {
  FinalizeArgs a{};
  a.partial = partial;
  a.nblk = nblk;
  a.M = M;
  a.C = C;
  a.gamma = gamma;
  a.invstd = invstd;
  a.dgamma = dgamma;
  a.dbeta = dbeta;
  a.beta_acc = beta_acc;
  a.coef = coef;
  if (fin_one_channel(C, nblk) && nblk > FIN_WIDE_ROWS) hipLaunchKernelGGL((bn_finalize_kernel<1, 1, 1024>), dim3(C), dim3(1024), 0, s, a);
  else if (fin_one_channel(C, nblk)) hipLaunchKernelGGL((bn_finalize_kernel<1, 1>), dim3(C), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((bn_finalize_kernel<1, 4>), dim3(C / 4), dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return 0;
}

int launch_bn_bwd_apply(int dtype, const void* g, const void* mask_src, const void* x, const float* mean,
                        const float* invstd, const float* coef, void* dx, int M, int C, hipStream_t s,
                        const uint8_t* relu_bits, float slope, QuantOut qo, const EcaGrad* eg) {
  MI355_TRY(check_c(dtype, C));
  const int V = 16 / (int)dtype_size(dtype);
  BwdApplyArgs a{};
  if (eg) a.eg = *eg;
  a.slope = slope;
  a.g = g;
  a.mask = mask_src;
  a.bits = relu_bits;
  a.x = x;
  a.mean = mean;
  a.invstd = invstd;
  a.coef = coef;
  a.dx = dx;
  a.nvec = (size_t)M * C / V;
  a.cvecs = C / V;
  a.C = C;
  const int blocks = elementwise_blocks(a.nvec, a.cvecs);
  const int mask = relu_bits ? (slope != 0.f ? 3 : 2) : mask_src ? 1 : 0;
  a.qo = qo;
  if (eg) {
    MI355_ARG(mask == 0 && !qo.q && eg->gate && eg->dpool && eg->hw > 0, "bn_bwd_apply: the ECA gradient source takes no mask / twin");
    if (dtype == MI355_F32) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 4>), dim3(blocks), dim3(256), 0, s, a);
    MI355_LAUNCH_CHECK();
    return 0;
  }
  if (qo.q) {
    MI355_ARG(dtype == MI355_BF16 && mask != 1 && mask != 3 && qo.scale && qo.amax, "bn_bwd_apply: the e4m3 twin needs bf16 and a ReLU bit mask (or none)");
    if (mask == 2 && qo.only) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 2, true, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (mask == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 2, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (qo.only) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 0, true, true>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 0, true>), dim3(blocks), dim3(256), 0, s, a);
    MI355_LAUNCH_CHECK();
    return 0;
  }
#define MI355_BN_BWD_APPLY(TT)                                                                                  \
  switch (mask) {                                                                                               \
    case 0: hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, 0>), dim3(blocks), dim3(256), 0, s, a); break;          \
    case 1: hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, 1>), dim3(blocks), dim3(256), 0, s, a); break;          \
    case 2: hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, 2>), dim3(blocks), dim3(256), 0, s, a); break;          \
    default: hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, 3>), dim3(blocks), dim3(256), 0, s, a); break;         \
  }
  if (dtype == MI355_F32) {
    MI355_BN_BWD_APPLY(float)
  } else {
    MI355_BN_BWD_APPLY(bf16_t)
  }
#undef MI355_BN_BWD_APPLY
  MI355_LAUNCH_CHECK();
  return 0;
}

static int stem_bwd_args(StemBwdArgs& a, int dtype, const void* dp, const uint8_t* idx, const uint8_t* bits, const void* y, const float* mean,
                         const float* invstd, int N, int H, int W, int C) {
  MI355_TRY(check_c(dtype, C));
  const int V = 16 / (int)dtype_size(dtype);
  (void)V;
  MI355_ARG(dp && idx && bits && y && H % 2 == 0 && W % 2 == 0 && C == SC, "stem_bwd: N=%d H=%d W=%d C=%d (64 channels)", N, H, W, C);
  MI355_ARG((unsigned long long)N * H * W * C * dtype_size(dtype) < 0xfffffff0ull, "stem_bwd: tensor beyond 32-bit offsets");
  a = StemBwdArgs{};
  a.dp = dp; a.idx = idx; a.bits = bits; a.y = y; a.mean = mean; a.invstd = invstd;
  a.N = N; a.Ho = H / 2; a.Wo = W / 2;
  return 0;
}

// BN-backward sums of a conv + BN + ReLU + maxpool3x3/2 stage from the POOLED gradient (H, W: the full resolution)
int launch_stem_bwd_reduce(int dtype, const void* dp, const uint8_t* idx, const uint8_t* bits, const void* y, const float* mean,
                           const float* invstd, float* partial, int* nblk_out, int N, int H, int W, int C, hipStream_t s) {
  StemBwdArgs a;
  MI355_TRY(stem_bwd_args(a, dtype, dp, idx, bits, y, mean, invstd, N, H, W, C));
  a.partial = partial;
  const int V = 16 / (int)dtype_size(dtype);
  const int rpp = 256 / (C / V);
  int nblk = cdiv(N * a.Ho * a.Wo, rpp * 4);
  const int cap = std::min(2 * MAXBLK, 2 * device_cus());  // one resident round of 2 workgroups per CU (<= bn_max_blocks() partial rows)
  if (nblk > cap) nblk = cap;
  *nblk_out = nblk;
  if (dtype == MI355_F32) hipLaunchKernelGGL(stem_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(stem_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return 0;
}

// ... and the gradient wrt the raw conv output, dx [N][H][W][C]
int launch_stem_bwd_apply(int dtype, const void* dp, const uint8_t* idx, const uint8_t* bits, const void* y, const float* mean,
                          const float* invstd, const float* coef, void* dx, int N, int H, int W, int C, hipStream_t s) {
  StemBwdArgs a;
  MI355_TRY(stem_bwd_args(a, dtype, dp, idx, bits, y, mean, invstd, N, H, W, C));
  a.coef = coef; a.dx = dx;
  const int V = 16 / (int)dtype_size(dtype);
  const int blocks = elementwise_blocks((size_t)N * a.Ho * a.Wo * (C / V), C / V);
  if (dtype == MI355_F32) hipLaunchKernelGGL(stem_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(stem_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // namespace mi355
