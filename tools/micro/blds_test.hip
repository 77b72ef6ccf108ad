// micro-test: buffer_load_dwordx4 ... offen lds — LDS placement (M0 + 16*lane) and out-of-range => zeros
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_srd(const void* p, unsigned bytes) {
  unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void blds16(i32x4 srd, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_dst) : "memory");
}
__global__ void k(const float* g, float* o, unsigned bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 1024; i += 256) ((float*)smem)[i] = -1.f;
  __syncthreads();
  i32x4 srd = make_srd(g, bytes);
  unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
  // lanes read a permuted source (lane^3), lane 5 of every wave is out of range
  unsigned voff = ((lane ^ 3) * 16 + wave * 1024) | (lane == 5 ? 0x80000000u : 0u);
  blds16(srd, voff, __builtin_amdgcn_readfirstlane(lds0 + wave * 1024));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int i = threadIdx.x; i < 1024; i += 256) o[i] = ((float*)smem)[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, g, o, 4096u);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    float exp = (l == 5) ? 0.f : (float)(w * 256 + (l ^ 3) * 4 + e);
    float got = r[w * 256 + l * 4 + e];
    if (got != exp) { if (bad < 8) printf("w%d l%d e%d got %f exp %f\n", w, l, e, got, exp); ++bad; }
  }
  printf("blds_test: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
