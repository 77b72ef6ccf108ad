// mfma_rate.hip — what the matrix pipe of gfx950 sustains per instruction form, operands in registers (no memory in the loop):
//   v_mfma_f32_16x16x32_bf16            the instruction of every bf16 conv kernel of this repo
//   v_mfma_f32_16x16x32_fp8_fp8         the e4m3 step's instruction (conv_igemm8.hip EB = 1)
//   v_mfma_f32_16x16x128_f8f6f4         e4m3 x e4m3 at 4x the K (no block scales)
//   v_mfma_scale_f32_16x16x128_f8f6f4   the same with E8M0 block scales (unit scales here)
// One wave per SIMD (4 per CU) or two, 8 independent accumulators per wave.  hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define LOOP8(INS)                                                                  \
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0; \
  for (int i = 0; i < iters; ++i) {                                                 \
    INS(c0) INS(c1) INS(c2) INS(c3) INS(c4) INS(c5) INS(c6) INS(c7)                 \
  }                                                                                 \
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];

__global__ void k_bf16(int iters, float* out) {
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
#define I_BF16(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  LOOP8(I_BF16)
}
__global__ void k_bf16_agpr(int iters, float* out) {  // accumulators in AGPRs, as in the generated kernels
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
#define I_BF16A(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  LOOP8(I_BF16A)
}
__global__ void k_bf16_agpr_b(int iters, float* out) {  // ... and the weight operand too (asm/po_gen.py)
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
#define I_BF16AB(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "v"(b));
  LOOP8(I_BF16AB)
}
__global__ void k_fp8(int iters, float* out) {
  i32x2 a = {(int)threadIdx.x, 1}, b = {4, (int)threadIdx.x};
#define I_FP8(c) asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  LOOP8(I_FP8)
}
__global__ void k_f8x128(int iters, float* out) {
  i32x8 a = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {4, 5, 6, 7, 8, 9, 10, (int)threadIdx.x};
#define I_F8X(c) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  LOOP8(I_F8X)
}
__global__ void k_f8x128s(int iters, float* out) {
  i32x8 a = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {4, 5, 6, 7, 8, 9, 10, (int)threadIdx.x};
  int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;  // E8M0 1.0
#define I_F8S(c) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
  LOOP8(I_F8S)
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_bf16_32(int iters, float* out) {
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
  f32x16 c0 = {0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
#define I_BF32(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    I_BF32(c0) I_BF32(c1) I_BF32(c2) I_BF32(c3) I_BF32(c0) I_BF32(c1) I_BF32(c2) I_BF32(c3)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <typename K>
static void run(const char* name, K kern, int K_, int wpc) {
  const int cus = 256, iters = 20000;
  float* out;
  CK(hipMalloc(&out, (size_t)cus * wpc * 64 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(cus), dim3(64 * wpc), 0, 0, iters, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  const double flop = 2.0 * 16 * 16 * K_ * 8.0 * iters * cus * wpc;
  const double cyc = best * 1e-3 * 2.4e9 / (8.0 * iters);  // cycles per MFMA per wave at the 2.4 GHz datasheet clock
  printf("%-38s %d waves/CU: %7.1f TFLOP/s  (%.1f cycles per instruction and wave at 2.4 GHz)\n", name, wpc, flop / (best * 1e-3) * 1e-12, cyc);
  CK(hipFree(out));
}

int main() {
  for (int wpc : {4, 8}) {
    run("v_mfma_f32_16x16x32_bf16", k_bf16, 32, wpc);
    run("v_mfma_f32_16x16x32_bf16, AGPR acc", k_bf16_agpr, 32, wpc);
    run("v_mfma_f32_16x16x32_bf16, AGPR acc + A", k_bf16_agpr_b, 32, wpc);
    run("v_mfma_f32_32x32x16_bf16 (x4 work)", k_bf16_32, 64, wpc);  // 2*32*32*16 = 2*16*16*64
    run("v_mfma_f32_16x16x32_fp8_fp8", k_fp8, 32, wpc);
    run("v_mfma_f32_16x16x128_f8f6f4", k_f8x128, 128, wpc);
    run("v_mfma_scale_f32_16x16x128_f8f6f4", k_f8x128s, 128, wpc);
  }
  return 0;
}
