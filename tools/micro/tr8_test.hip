// micro-test: lane mapping of ds_read_b64_tr_b8 on gfx950 (the byte form of the transposed LDS read; only the 16-bit form is
// described in the guides).  LDS holds a [64 rows][64 bytes] image with byte (r, c) = r * 64 + c... too wide for a byte: we store
// r in one pass and c in a second pass and print, per lane, the (row, col) of each of its 8 result bytes for the address pattern
// "lane l supplies row_of(l), col_of(l)" under the hypothesis below.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/tr8_test.hip -o gpurun_out/tr8_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef v2i __attribute__((address_space(3))) * lds_v2i_ptr;
__global__ void k(unsigned char* out_r, unsigned char* out_c, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char img[2][64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) {
    img[0][i] = (unsigned char)(i / 64);
    img[1][i] = (unsigned char)(i % 64);
  }
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, w = lane & 15;
  // hypothesis A (by analogy with tr_b16): per 16-lane group a block of 8 rows x 16 byte-columns; lane 2q + p of the group supplies
  // the address of row q, columns 8p .. 8p+7; lane i of the group receives column i of the 8 rows.
  // mode 0: addresses per hypothesis A, group g reads block rows 8g.., cols 0..15
  // mode 1: every lane supplies row (lane & 15), col 0 (to see which lanes' addresses are used for which output byte)
  int row, col;
  if (mode == 0) { row = 8 * g + (w >> 1); col = 8 * (w & 1); }
  else { row = w + 16 * g; col = 8 * (g & 1); }
  for (int pass = 0; pass < 2; ++pass) {
    const v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr)(&img[pass][row * 64 + col]));
    unsigned char* o = pass == 0 ? out_r : out_c;
    for (int e = 0; e < 8; ++e) o[lane * 8 + e] = (unsigned char)((e < 4 ? (unsigned)v[0] >> (8 * e) : (unsigned)v[1] >> (8 * (e - 4))) & 0xff);
  }
}
int main() {
  unsigned char *r, *c;
  hipMalloc(&r, 512); hipMalloc(&c, 512);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, r, c, mode);
    unsigned char hr[512], hc[512];
    hipMemcpy(hr, r, 512, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 512, hipMemcpyDeviceToHost);
    printf("mode %d: lane: (row,col) of result bytes 0..7\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int e = 0; e < 8; ++e) printf(" (%2d,%2d)", hr[l * 8 + e], hc[l * 8 + e]);
      printf("\n");
    }
  }
  return 0;
}
