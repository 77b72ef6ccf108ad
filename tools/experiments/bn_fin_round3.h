// EXPERIMENT RECORD (round 3; removed from the product library in round 4) — BatchNorm finalize by the last-arriving workgroup of the
// launch that sums the statistics.  Measured: 20.4 -> 33.3 ms per step with it on (agent-scope ticket atomics on one address run at
// ~4 M/s across the 8 XCDs: ~120 us per launch); a K-arriver variant 32.9 ms; a flag-polling variant without atomics 20.8-21.0 ms
// against 20.0 (profiles/README.md, round 3).  The call sites it had (bn.hip, conv_igemm.hip, conv_igemm8.hip, resnet_exec.cpp) are in
// the history before the commit that moved this file here; it does not build on its own.
// bn_fin.h — BatchNorm finalize carried out by the LAST-ARRIVING workgroup of the launch that produced the partial rows.
//
// The conv epilogues (conv_igemm.hip / conv_igemm8.hip, STATS 1 / 2) and the standalone reduce kernel (bn.hip) leave one row of
// per-channel partial sums per workgroup; a separate bn_finalize launch then added the <= 512 rows (7 us + a kernel boundary, 106
// times per training step).  Here every workgroup of a channel group
//   1. writes its row with write-through (sc1) stores and drains them,
//   2. counts itself on the group's counter (relaxed agent-scope atomic),
// and the workgroup that draws the last ticket issues ONE agent-scope acquire, adds the rows in ROW order (fp64, the fixed-shape
// tree of bn_finalize_kernel: the result does not depend on which workgroup arrives last — bitwise reproducible), writes
// scale / shift / saved statistics / running statistics (forward) or dgamma / dbeta / the three backward coefficients, and
// re-arms the counter.  Nobody spins and no fence runs on a cache that still feeds the main loop.
#pragma once
#include "common.h"

namespace mi355 {

// one 256-thread slab finalizes 4 channels per pass: 64 slices x 4 channels, <= 8 rows per thread, then an LDS tree
__device__ __forceinline__ void bn_fin_channels(const BnFinArgs& f, const float* __restrict__ partial, int nblk, int C, int c_begin, int c_count,
                                                double* red /* LDS: [slabs][2][64][4] doubles */, int tid, int nthreads) {
  const int slab = tid >> 8, nslab = nthreads >> 8, t = tid & 255;
  const int cl = t & 3, sl = t >> 2;
  double* r0 = red + slab * 512;
  double* r1 = r0 + 256;
  for (int cq = slab * 4; cq < c_count; cq += nslab * 4) {  // (c_count is a multiple of 4 * nslab for every launch of the network)
    const int c = c_begin + cq + cl;
    const bool live = cq + cl < c_count;
    double a = 0.0, b = 0.0;
    if (live) {
      for (int k0 = sl; k0 < nblk; k0 += 64 * 8) {
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 64 * u;
          const bool in = k < nblk;
          va[u] = in ? partial[((size_t)k * 2 + 0) * C + c] : 0.f;
          vb[u] = in ? partial[((size_t)k * 2 + 1) * C + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a += (double)va[u];
          b += (double)vb[u];
        }
      }
    }
    r0[sl * 4 + cl] = a;
    r1[sl * 4 + cl] = b;
    __syncthreads();
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      if (sl < s) {
        r0[sl * 4 + cl] += r0[(sl + s) * 4 + cl];
        r1[sl * 4 + cl] += r1[(sl + s) * 4 + cl];
      }
      __syncthreads();
    }
    if (sl == 0 && live) {
      const double s1 = r0[cl], s2 = r1[cl];
      const double M = (double)f.M;
      if (f.mode == 1) {
        const double dm = s1 / M;  // mean of (x - pivot)
        const double mean = (f.pivot ? (double)f.pivot[c] : 0.0) + dm;
        double var = s2 / M - dm * dm;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float meanf = (float)mean;
        f.save_mean[c] = meanf;
        f.save_invstd[c] = invstd;
        const float sc = f.gamma[c] * invstd;
        f.scale[c] = sc;
        f.shift[c] = f.beta[c] - meanf * sc;
        if (f.running_mean) {
          const double unb = f.M > 1 ? var * M / (M - 1.0) : var;
          f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * meanf;
          f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
        }
      } else {
        const float db = (float)s1, dg = (float)s2;
        f.dbeta[c] = (f.beta_acc != 0.f ? f.beta_acc * f.dbeta[c] : 0.f) + db;
        f.dgamma[c] = (f.beta_acc != 0.f ? f.beta_acc * f.dgamma[c] : 0.f) + dg;
        f.coef[c] = f.gamma[c] * f.invstd[c];
        f.coef[C + c] = (float)(s1 / M);
        f.coef[2 * C + c] = (float)(s2 / M);
      }
    }
    __syncthreads();
  }
}

// write-through store of one float (the row must be readable by a workgroup behind another XCD's L2)
__device__ __forceinline__ void store_wt(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Called by ALL threads of the workgroup after its row stores (store_wt) have been issued.  `group`: the channel group this
// workgroup flushed (its counter), `members`: workgroups in the group, `rows`: partial rows to add, channels [c_begin, c_begin + c_count).
// lds: >= (nthreads / 256) * 4 KiB + 16 bytes of LDS nobody else uses any more.
__device__ __forceinline__ void bn_fin_last_arriver(const BnFinArgs& f, const float* partial, int rows, int C, int group, int members, int c_begin,
                                                    int c_count, char* lds, int tid, int nthreads) {
  int* ticket = reinterpret_cast<int*>(lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's row stores have been acknowledged by memory
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(f.counters + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == (unsigned)(members - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(f.counters + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch on this stream
    }
    *ticket = last ? 1 : 0;
  }
  __syncthreads();
  if (*ticket == 0) return;
  bn_fin_channels(f, partial, rows, C, c_begin, c_count, reinterpret_cast<double*>(lds + 16), tid, nthreads);
}

}  // namespace mi355
