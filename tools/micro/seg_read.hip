// seg_read.hip — how fast does HBM stream when a wave-instruction's 64 x 16 bytes are laid out as the MFMA epilogues of the generated
// conv kernels lay them out (16 pixel rows x 64 contiguous bytes, the other 64-byte half of each 128-byte line fetched by a LATER
// instruction), against full-line layouts?  Reads a [M][N] bf16 tensor once per launch from cold caches (pool of tensors > MALL).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/seg_read.hip -o /tmp/seg_read && /tmp/seg_read
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: linear (lane * 16 bytes, 1 KiB contiguous per wave instruction)
// MODE 1: 16 rows x 64 bytes per instruction; the two halves of a line by consecutive instructions
// MODE 2: 16 rows x 64 bytes; all first halves of a 64-row x 512-byte block first, then all second halves (as the epilogue items do)
// MODE 3: 8 rows x 128 bytes per instruction
// MODE 4: 4 rows x 256 bytes per instruction
// A workgroup (4 waves) walks 64-row tiles of a 512-byte column stripe (wave w: its 128 bytes), ILP loads in flight per wave.
template <int MODE, int WRITE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, char* __restrict__ dst, unsigned* sink, int M, int pitch, int tiles_per_wg, int nct) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int x = blockIdx.x, xcd = x & 7, l = x >> 3, ct = l % nct, g = (l / nct) * 8 + xcd;
  u32x4 acc = {0, 0, 0, 0};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long row0 = ((long)g * tiles_per_wg + t) * 64;
    if (row0 >= M) break;
    const char* base = src + row0 * pitch + ct * 512;
    char* dbase = dst + row0 * pitch + ct * 512;
    u32x4 v[8];
    long off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {        // the wave's 8 KiB of the tile as 8 linear KiB (tile stored contiguously: not a stripe — the yardstick)
        off[i] = ((long)(w * 8 + i) * 1024 + lane * 16);
        off[i] = (off[i] / 512) * pitch + off[i] % 512;  // row = off / 512 of the stripe: 32 lanes per row
      } else if (MODE == 1) { // item i = (fragment m = i / 2, half p = i % 2): rows m*16 + (lane & 15), bytes w*128 + p*64 + (lane >> 4)*16
        off[i] = (long)((i >> 1) * 16 + (lane & 15)) * pitch + w * 128 + (i & 1) * 64 + (lane >> 4) * 16;
      } else if (MODE == 2) { // all p = 0 first
        off[i] = (long)((i & 3) * 16 + (lane & 15)) * pitch + w * 128 + (i >> 2) * 64 + (lane >> 4) * 16;
      } else if (MODE == 3) { // 8 rows x 128 bytes: row i*8 + (lane >> 3), bytes w*128 + (lane & 7)*16
        off[i] = (long)(i * 8 + (lane >> 3)) * pitch + w * 128 + (lane & 7) * 16;
      } else {                // 4 rows x 256 bytes... of a 512-byte stripe shared by 2 waves: rows (w >> 1)*32 + i*4 + (lane >> 4), bytes (w & 1)*256 + (lane & 15)*16
        off[i] = (long)((w >> 1) * 32 + i * 4 + (lane >> 4)) * pitch + (w & 1) * 256 + (lane & 15) * 16;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc += v[i];
      if (WRITE) __builtin_nontemporal_store(v[i] + acc, reinterpret_cast<u32x4*>(dbase + off[i]));
    }
  }
  if (acc.x == 0x12345678u && acc.y == 1) sink[0] = acc.z + acc.w;
}

template <int MODE, int WRITE>
void run(const char* name, std::vector<char*>& pool, std::vector<char*>& dpool, unsigned* sink, int M, int N) {
  const int pitch = N * 2, nct = pitch / 512;
  const int groups = 256 / nct, tiles = (M + 63) / 64, tpw = (tiles + groups - 1) / groups;
  const int grid = (groups + 7) / 8 * 8 * nct;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < 25; ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, WRITE>), dim3(grid), dim3(256), 0, 0, pool[it % pool.size()], dpool[it % pool.size()], sink, M, pitch, tpw, nct);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 5) ts.push_back(ms * 1000.f);
  }
  std::sort(ts.begin(), ts.end());
  const double bytes = (double)M * pitch * (1 + WRITE);
  printf("%-46s M %7d N %5d: median %7.1f us  %5.2f TB/s\n", name, M, N, ts[ts.size() / 2], bytes / ts[ts.size() / 2] * 1e-6);
}

int main() {
  unsigned* sink;
  CK(hipMalloc((void**)&sink, 64));
  for (auto mn : {std::pair<int, int>{50176, 1024}, {802816, 256}, {200704, 512}}) {
    const int M = mn.first, N = mn.second;
    const size_t bytes = (size_t)M * N * 2;
    const int POOL = (int)(1.5e9 / bytes) + 2;
    std::vector<char*> pool(POOL), dpool(POOL);
    for (auto& p : pool) {
      CK(hipMalloc((void**)&p, bytes));
      CK(hipMemset(p, 1, bytes));
    }
    for (auto& p : dpool) CK(hipMalloc((void**)&p, bytes));
    run<0, 0>("read linear KiB per instruction", pool, dpool, sink, M, N);
    run<1, 0>("read 16 rows x 64 B, halves adjacent", pool, dpool, sink, M, N);
    run<2, 0>("read 16 rows x 64 B, halves 4 instructions apart", pool, dpool, sink, M, N);
    run<3, 0>("read 8 rows x 128 B", pool, dpool, sink, M, N);
    run<4, 0>("read 4 rows x 256 B", pool, dpool, sink, M, N);
    run<1, 1>("copy 16 rows x 64 B, halves adjacent", pool, dpool, sink, M, N);
    run<2, 1>("copy 16 rows x 64 B, halves 4 apart", pool, dpool, sink, M, N);
    run<3, 1>("copy 8 rows x 128 B", pool, dpool, sink, M, N);
    run<4, 1>("copy 4 rows x 256 B", pool, dpool, sink, M, N);
    for (auto& p : pool) CK(hipFree(p));
    for (auto& p : dpool) CK(hipFree(p));
  }
  return 0;
}
