// dconv_bench.cpp — standalone timing harness for the generated direct-conv kernels (asm/dconv_gen.py), tuning only.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/dconv_bench.cpp -o /tmp/dconv_bench
//   /tmp/dconv_bench <file.hsaco> <kernel> <file.tbl> H W IPT Cin NCOLS N [nchunks] [iters]
// Launches the kernel on random bf16 data from a pool of inputs larger than the Infinity Cache and prints the median /
// minimum time per launch.  nchunks overrides the reduction length (results are then wrong; the time per chunk is the point).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 10) {
    fprintf(stderr, "usage: %s hsaco kernel tbl H W IPT Cin NCOLS N [nchunks] [iters]\n", argv[0]);
    return 2;
  }
  const char* hsaco = argv[1];
  const char* kname = argv[2];
  const char* tblf = argv[3];
  const int H = atoi(argv[4]), W = atoi(argv[5]), IPT = atoi(argv[6]), Cin = atoi(argv[7]), NCOLS = atoi(argv[8]), N = atoi(argv[9]);
  const int nchunks = argc > 10 ? atoi(argv[10]) : Cin / 64;
  const int iters = argc > 11 ? atoi(argv[11]) : 40;
  const int BNC = argc > 12 ? atoi(argv[12]) : 256;  // columns per workgroup
  hipModule_t mod;
  hipFunction_t fn;
  CK(hipModuleLoad(&mod, hsaco));
  CK(hipModuleGetFunction(&fn, mod, kname));
  unsigned tbl[768];
  memset(tbl, 0, sizeof(tbl));
  FILE* f = fopen(tblf, "rb");
  if (!f || fread(tbl, 4, 768, f) < 128) {
    fprintf(stderr, "cannot read %s\n", tblf);
    return 1;
  }
  fclose(f);
  const int POOL = 12;
  const size_t in_elems = (size_t)N * H * W * Cin, wt_elems = (size_t)NCOLS * 9 * Cin + 4096, out_elems = (size_t)N * H * W * NCOLS;
  std::vector<unsigned short> h(std::max(in_elems, wt_elems));
  unsigned short* d_in[POOL];
  srand(1);
  for (int p = 0; p < POOL; ++p) {
    for (size_t i = 0; i < in_elems; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff)) ^ (unsigned short)((rand() & 1) << 15);  // random sign, magnitude ~[0.5, 2)
    CK(hipMalloc((void**)&d_in[p], in_elems * 2));
    CK(hipMemcpy(d_in[p], h.data(), in_elems * 2, hipMemcpyHostToDevice));
  }
  unsigned short *d_wt, *d_out;
  float* d_stat;
  unsigned* d_tbl;
  for (size_t i = 0; i < wt_elems; ++i) h[i] = (unsigned short)(0x3800 + (rand() & 0x3ff)) ^ (unsigned short)((rand() & 1) << 15);
  CK(hipMalloc((void**)&d_wt, wt_elems * 2));
  CK(hipMemcpy(d_wt, h.data(), wt_elems * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&d_out, out_elems * 2));
  CK(hipMalloc((void**)&d_stat, (size_t)(IPT < 0 ? N * -IPT : N / IPT) * 2 * NCOLS * 4 + 4096));
  CK(hipMalloc((void**)&d_tbl, 512));
  CK(hipMemcpy(d_tbl, tbl, 512, hipMemcpyHostToDevice));
  struct __attribute__((packed)) KArgs {
    const void* in;
    const void* wt;
    void* out;
    float* stat;
    const void* p4[4];
    const void* rsvd;
    unsigned wtap_off[9];
    unsigned nchunks;
    unsigned pad[4];
    unsigned table[768];
  } k;
  memset(&k, 0, sizeof(k));
  k.wt = d_wt;
  k.out = d_out;
  k.stat = d_stat;
  // the BN-backward variants read y / mask / mean / invstd: any readable memory of the right size will do for timing
  k.p4[0] = d_out; k.p4[1] = d_out; k.p4[2] = d_stat; k.p4[3] = d_stat;
  memcpy(k.table, tbl, sizeof(k.table));
  for (int t = 0; t < 9; ++t) k.wtap_off[t] = (unsigned)(t * Cin * 2);
  k.nchunks = (unsigned)nchunks;
  // pointwise kernels (asm/pw_gen.py): kernel name "pw_*", IPT = rows per unit; H*W*N = pixels
  const bool pw = strncmp(kname, "pw_", 3) == 0;
  struct __attribute__((packed)) PwArgs {
    const void* in;
    const void* wt;
    void* out;
    float* stat;
    const void* rsvd[6];
    unsigned units, upw, mtiles, pad0;
    unsigned pad[8];
    unsigned table[128];
  } pk;
  unsigned pw_grid = 0;
  if (pw) {
    memset(&pk, 0, sizeof(pk));
    pk.wt = d_wt; pk.out = d_out; pk.stat = d_stat;
    const long M = (long)N * H * W;
    pk.mtiles = (unsigned)(M / IPT);
    pk.units = pk.mtiles * (unsigned)(NCOLS / 256);
    pk.upw = (pk.units + 255) / 256;
    pw_grid = (pk.units + pk.upw - 1) / pk.upw;
    memcpy(pk.table, tbl, sizeof(pk.table));
  }
  size_t ksize = pw ? sizeof(pk) : sizeof(k);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < iters + 5; ++it) {
    k.in = d_in[it % POOL];
    pk.in = d_in[it % POOL];
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, pw ? (void*)&pk : (void*)&k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
    CK(hipEventRecord(e0, 0));
    if (pw) CK(hipModuleLaunchKernel(fn, pw_grid, 1, 1, 256, 1, 1, 0, 0, nullptr, extra));
    else CK(hipModuleLaunchKernel(fn, (unsigned)(IPT < 0 ? N * -IPT : N / IPT), (unsigned)(NCOLS / BNC), 1, 256, 1, 1, 0, 0, nullptr, extra));  // IPT < 0: -IPT tiles per image
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 5) ts.push_back(ms * 1000.f);
  }
  std::sort(ts.begin(), ts.end());
  const double fl = 2.0 * N * H * W * (double)NCOLS * 64.0 * nchunks * (strncmp(kname, "pw_", 3) == 0 ? 1 : 9);
  printf("%-28s nchunks %d: median %7.1f us  min %7.1f us  %7.1f TF/s (median)\n", kname, nchunks, ts[ts.size() / 2], ts[0], fl / ts[ts.size() / 2] * 1e-6);
  return 0;
}
