// micro-test: what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report per byte for the access patterns of this library.
// MI355X_MICROARCH.md: "on gfx950 FETCH_SIZE reports exactly 1/2 of a wide coalesced 16-B-per-lane streaming read; other access
// widths are uncalibrated — calibrate on a known byte count in your own access pattern".  Each kernel below reads (writes) every
// byte of a 512 MiB buffer exactly once; the launch name says the pattern:
//   seg<S>: a wave-instruction covers 1 KiB as 1024/S segments of S contiguous bytes, segments 2 KiB apart (the conv epilogues read
//           the shortcut addend / the BN tensors and write their output like that: S = 128 for 64-column wave tiles, 256 for 128)
//   seg1024 = the plain streaming pattern (the BN kernels, LDS-DMA staging of dense rows)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o gpurun_out/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S, bool WRITE>
__global__ __launch_bounds__(256) void seg(uint4* buf, size_t nbytes, uint4* sink) {
  // the buffer is viewed as rows of 2 KiB; a "column block" of S bytes x (1024/S) consecutive rows is one wave-instruction
  constexpr int LPS = S / 16;          // lanes per segment
  constexpr int RPI = 64 / LPS;        // rows per wave-instruction
  constexpr int CB = 2048 / S;         // column blocks per row
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t units = nbytes / 1024;  // wave-instructions in total
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t u = wave; u < units; u += nwaves) {
    const size_t rowgrp = u / CB, cb = u % CB;
    const size_t row = rowgrp * RPI + lane / LPS;
    const size_t off = row * 2048 + cb * S + (lane % LPS) * 16;
    if (WRITE) {
      buf[off / 16] = make_uint4((unsigned)u, lane, 0, 0);
    } else {
      const uint4 v = buf[off / 16];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  if (!WRITE && acc.x == 0x12345678u) sink[0] = acc;
}
int main() {
  const size_t n = 512ull << 20;
  uint4 *buf, *sink;
  hipMalloc(&buf, n); hipMalloc(&sink, 64);
  hipMemset(buf, 1, n);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((seg<1024, false>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<256, false>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<128, false>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<64, false>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<1024, true>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<256, true>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<128, true>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
    hipLaunchKernelGGL((seg<64, true>), dim3(2048), dim3(256), 0, 0, buf, n, sink);
  }
  hipDeviceSynchronize();
  printf("done %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
