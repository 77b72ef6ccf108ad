// dconv.cpp — host side of the generated direct 3x3 / stride-1 convolution kernels (asm/dconv_gen.py): the code object is
// embedded in this library, loaded per device with hipModuleLoadData, and launched with the IgemmArgs contract of
// launch_igemm() (conv forward / data gradient under `model(data)` / `loss.backward()`,
// /root/reference/sota_imagenet/callbacks.py:316-317).
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace mi355 {

namespace {

struct DconvVariant {
  const char* name;
  int H, W, IPT, TPI, BN, Cin, NCOLS, stats, s2d, bnin, fp8, lds, kernarg;  // IPT images per tile, or TPI tiles per image; BN columns per workgroup;
                                                                 // s2d: the data gradient of a stride-2 3x3 (H x W = the dy image, four classes);
                                                                 // bnin: BatchNorm + ReLU of the input in the operand path (IgemmArgs::bn_in);
                                                                 // fp8: e4m3 operands (one byte per element), 128-channel chunks, K = 128 MFMA
  unsigned table[3 * 4 * 64];  // [tile class][wave] LDS-DMA piece tables (asm/dconv_gen.py tables())
};

const DconvVariant g_variants[] = {
#include "build/asm/dconv_meta.inc"
};
constexpr int NVAR = (int)(sizeof(g_variants) / sizeof(g_variants[0]));

// transform tables of the bnin kernels (asm/dconv_gen.py ttables()): [tile class][wave][64]
struct DconvTT {
  const char* name;
  unsigned table[3 * 4 * 64];
};
const DconvTT g_tt[] = {
#include "build/asm/dconv_tt.inc"
};
constexpr int NTT = (int)(sizeof(g_tt) / sizeof(g_tt[0]));

// persistent pointwise kernels (asm/pw_gen.py)
struct PwVariant {
  const char* name;
  int K, N, stats, rows, lds, kernarg;
  unsigned table[4 * 32];
};
const PwVariant g_pw[] = {
#include "build/asm/pw_meta.inc"
};
constexpr int NPW = (int)(sizeof(g_pw) / sizeof(g_pw[0]));

// long-reduction pointwise kernels (asm/pk_gen.py): tiles of W pixels x BN (256 | 128) columns
struct PkVariant {
  const char* name;
  int W, K, N, BN, stats, lds, kernarg;
};
const PkVariant g_pk[] = {
#include "build/asm/pk_meta.inc"
};
constexpr int NPK = (int)(sizeof(g_pk) / sizeof(g_pk[0]));

// 3x3 weight-gradient kernels (asm/wg_gen.py)
struct WgVariant {
  const char* name;
  int H, W, C, CO, tn, ti, lds, kernarg;  // tn / ti: tiles per image as a fraction
};
const WgVariant g_wg[] = {
#include "build/asm/wg_meta.inc"
};
constexpr int NWG = (int)(sizeof(g_wg) / sizeof(g_wg[0]));

// 1x1 weight-gradient kernels (asm/wg1_gen.py): 64 XP ci x 64 DP co per workgroup, tiles of 64 pixels
struct Wg1Variant {
  const char* name;
  int C, CO, XP, DP, lds, kernarg;
};
const Wg1Variant g_wg1[] = {
#include "build/asm/wg1_meta.inc"
};
constexpr int NWG1 = (int)(sizeof(g_wg1) / sizeof(g_wg1[0]));

// output-heavy pointwise kernels with the weights resident in AGPRs (asm/po_gen.py): persistent workgroups, TP-pixel tiles x BN columns;
// WM = 2 (the 64-column forms): the two waves that share columns leave a statistics row each -> 2 rows per workgroup
struct PoVariant {
  const char* name;
  int K, BN, stats, add, TP, WM, bnin, lds, kernarg;  // bnin: BatchNorm + ReLU of the input in the operand path (IgemmArgs::bn_in)
};
const PoVariant g_po[] = {
#include "build/asm/po_meta.inc"
};
constexpr int NPO = (int)(sizeof(g_po) / sizeof(g_po[0]));

alignas(4096) const unsigned char g_blob[] = {
#include "build/asm/dconv_blob.inc"
};

struct DevState {
  bool ok = false;
  int attempts = 0;
  hipModule_t mod = nullptr;
  hipFunction_t fn[NVAR] = {};
  hipFunction_t pw[NPW] = {};
  hipFunction_t wg[NWG] = {};
  hipFunction_t pk[NPK] = {};
  hipFunction_t wg1[NWG1] = {};
  hipFunction_t po[NPO] = {};
};
DevState g_dev[64];
std::mutex g_mu;

// loads the module on the current device (once); false with the error set when the runtime refuses it.  A refusal is remembered per
// device ("failed"): the *_legal() checks below then keep every launch on the implicit-GEMM kernels instead of failing it, and one line
// on stderr says so.  A transient refusal (out of memory at load time) is retried a few times before it is taken as final.
bool dev_state(DevState** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    set_error("dconv: no current device");
    return false;
  }
  std::lock_guard<std::mutex> lock(g_mu);
  DevState& d = g_dev[dev];
  if (!d.ok && d.attempts < 3) {
    ++d.attempts;
    hipError_t e = hipSuccess;
    const char* what = "hipModuleLoadData";
    if (d.mod == nullptr) e = hipModuleLoadData(&d.mod, g_blob);
    auto get = [&](hipFunction_t* f, const char* name) {
      if (e != hipSuccess) return;
      e = hipModuleGetFunction(f, d.mod, name);
      if (e != hipSuccess) what = name;
    };
    for (int i = 0; i < NVAR; ++i) get(&d.fn[i], g_variants[i].name);
    for (int i = 0; i < NPW; ++i) get(&d.pw[i], g_pw[i].name);
    for (int i = 0; i < NWG; ++i) get(&d.wg[i], g_wg[i].name);
    for (int i = 0; i < NPK; ++i) get(&d.pk[i], g_pk[i].name);
    for (int i = 0; i < NWG1; ++i) get(&d.wg1[i], g_wg1[i].name);
    for (int i = 0; i < NPO; ++i) get(&d.po[i], g_po[i].name);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("dconv: %s -> %s", what, hipGetErrorString(e));
      if (e != hipErrorOutOfMemory || d.attempts >= 3) {
        d.attempts = 3;
        fprintf(stderr, "mi355rn: the generated gfx950 kernels are unavailable on device %d (%s -> %s); every convolution stays on the implicit-GEMM kernels\n",
                dev, what, hipGetErrorString(e));
      }
      return false;
    }
    d.ok = true;
  }
  if (!d.ok) {
    set_error("dconv: module load failed earlier on this device");
    return false;
  }
  *out = &d;
  return true;
}

// the generated kernels can be launched on the current device (loads the module on first use)
bool module_ok() {
  DevState* d = nullptr;
  return dev_state(&d);
}

// the 9 taps of a stride-1 3x3 class as wtap[(dh + 1)*3 + (dw + 1)], false when the class is not that pattern
bool tap_table(const TapClass& c, int wtap[9]) {
  if (c.ntaps != 9 || c.ph != 0 || c.pw != 0) return false;
  bool seen[9] = {};
  for (int t = 0; t < 9; ++t) {
    const Tap& tp = c.taps[t];
    if (tp.dh < -1 || tp.dh > 1 || tp.dw < -1 || tp.dw > 1 || tp.wtap < 0 || tp.wtap >= 9) return false;
    const int k = (tp.dh + 1) * 3 + (tp.dw + 1);
    if (seen[k]) return false;
    seen[k] = true;
    wtap[k] = tp.wtap;
  }
  return true;
}

// the stride-2 data gradient's classes in the kernel's order ((ph, pw) = (1, 1), (1, 0), (0, 1), (0, 0); asm/dconv_gen.py Gen.S2D_CLASSES) and, per class,
// its taps (dh, dw) in {0, 1}^2: wtap[slot] = the weight tap of kernel slot `slot` (4 + 2 + 2 + 1 slots); false when the launch is not that pattern
bool s2d_tap_table(const IgemmArgs& a, int nclass, int wtap[9]) {
  static const int kcls[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
  static const int ktaps[4][4][2] = {{{0, 0}, {0, 1}, {1, 0}, {1, 1}}, {{0, 0}, {1, 0}, {-1, -1}, {-1, -1}}, {{0, 0}, {0, 1}, {-1, -1}, {-1, -1}}, {{0, 0}, {-1, -1}, {-1, -1}, {-1, -1}}};
  static const int kn[4] = {4, 2, 2, 1};
  if (nclass != 4) return false;
  int slot = 0;
  for (int k = 0; k < 4; ++k) {
    const TapClass* c = nullptr;
    for (int i = 0; i < 4; ++i)
      if (a.cls[i].ph == kcls[k][0] && a.cls[i].pw == kcls[k][1]) c = &a.cls[i];
    if (c == nullptr || c->ntaps != kn[k]) return false;
    for (int j = 0; j < kn[k]; ++j, ++slot) {
      int w = -1;
      for (int t = 0; t < c->ntaps; ++t)
        if (c->taps[t].dh == ktaps[k][j][0] && c->taps[t].dw == ktaps[k][j][1]) w = c->taps[t].wtap;
      if (w < 0 || w >= 9) return false;
      wtap[slot] = w;
    }
  }
  return slot == 9;
}

int find_variant_s2d(const IgemmArgs& a, int nclass, int stats) {
  if (nclass != 4 || a.IS != 1 || a.OS != 2 || a.pair_delta != 0 || a.wtaps != 9) return -1;
  if (a.Hsub != a.Hin || a.Wsub != a.Win || a.Hout != 2 * a.Hin || a.Wout != 2 * a.Win) return -1;
  if (a.pix_stride != a.Ck || a.addend != nullptr || a.q_scale_in != nullptr || a.q_scale_wt != nullptr) return -1;
  if (stats == 1) return -1;
  int wtap[9];
  if (!s2d_tap_table(a, nclass, wtap)) return -1;
  for (int i = 0; i < NVAR; ++i) {
    const DconvVariant& v = g_variants[i];
    if (v.s2d && v.H == a.Hin && v.W == a.Win && v.Cin == a.Ck && v.NCOLS == a.Ncols && v.stats == stats && a.N % v.IPT == 0) return i;
  }
  return -1;
}

int find_variant(const IgemmArgs& a, int nclass, int stats, int fp8 = 0) {
  if (nclass == 4) return fp8 ? -1 : find_variant_s2d(a, nclass, stats);
  if (nclass != 1 || a.IS != 1 || a.OS != 1 || a.pair_delta != 0 || a.wtaps != 9) return -1;
  if (a.Hsub != a.Hin || a.Wsub != a.Win || a.Hout != a.Hin || a.Wout != a.Win) return -1;
  if (a.pix_stride != a.Ck || a.addend != nullptr) return -1;  // (sk_ws is optional scratch: not needed here)
  if (!fp8 && (a.q_scale_in != nullptr || a.q_scale_wt != nullptr)) return -1;
  if (fp8 && ((a.q_scale_in == nullptr) != (a.q_scale_wt == nullptr) || a.bn_in != nullptr)) return -1;
  int wtap[9];
  if (!tap_table(a.cls[0], wtap)) return -1;
  const int bnin = a.bn_in != nullptr ? 1 : 0;
  if (bnin && (a.bn_in_a == nullptr || a.bn_in_bits == nullptr)) return -1;
  for (int i = 0; i < NVAR; ++i) {
    const DconvVariant& v = g_variants[i];
    if (v.bnin != bnin || v.fp8 != fp8) continue;
    if (!v.s2d && v.H == a.Hin && v.W == a.Win && v.Cin == a.Ck && v.NCOLS == a.Ncols && v.stats == stats && a.N % v.IPT == 0) return i;
  }
  return -1;
}

// the persistent pointwise kernel that can run this launch (1x1, stride 1, no addend, statistics 0 / 1), or -1
int find_pw(const IgemmArgs& a, int nclass, int stats) {
  if (nclass != 1 || a.IS != 1 || a.OS != 1 || a.pair_delta != 0 || a.wtaps != 1 || a.cls[0].ntaps != 1) return -1;
  if (a.cls[0].taps[0].dh != 0 || a.cls[0].taps[0].dw != 0 || a.cls[0].taps[0].wtap != 0 || a.cls[0].ph != 0 || a.cls[0].pw != 0) return -1;
  if (a.Hsub != a.Hin || a.Wsub != a.Win || a.Hout != a.Hin || a.Wout != a.Win) return -1;
  if (a.pix_stride != a.Ck || a.addend != nullptr || a.q_scale_in != nullptr || a.q_scale_wt != nullptr) return -1;
  if (stats >= 2) return -1;
  const long M = (long)a.N * a.Hin * a.Win;
  for (int i = 0; i < NPW; ++i) {
    const PwVariant& v = g_pw[i];
    if (v.K == a.Ck && v.N == a.Ncols && v.stats == stats && M % v.rows == 0 && M / v.rows < (1 << 20)) return i;
  }
  return -1;
}

// the generated 3x3 / stride-1 weight-gradient kernel of this launch, or -1
int find_wg(int dtype, const WgradArgs& a) {
  if (dtype != MI355_BF16 || a.ntaps != 9 || a.wtaps != 9 || a.IS != 1 || a.pair_delta != 0) return -1;
  if (a.Ho != a.Hin || a.Wo != a.Win || a.pix_stride != a.Ck) return -1;
  for (int t = 0; t < 9; ++t)
    if (a.taps[t].dh != t / 3 - 1 || a.taps[t].dw != t % 3 - 1 || a.taps[t].wtap != t) return -1;
  for (int i = 0; i < NWG; ++i) {
    const WgVariant& v = g_wg[i];
    if (v.H == a.Hin && v.W == a.Win && v.C == a.Ck && v.CO == a.Cout && (a.N * v.tn) % v.ti == 0) return i;
  }
  return -1;
}

// the generated 1x1 / stride-1 weight-gradient kernel of this launch, or -1
int find_wg1(int dtype, const WgradArgs& a) {
  if (dtype != MI355_BF16 || a.ntaps != 1 || a.wtaps != 1 || a.IS != 1 || a.pair_delta != 0) return -1;
  if (a.Ho != a.Hin || a.Wo != a.Win || a.pix_stride != a.Ck || a.taps[0].dh != 0 || a.taps[0].dw != 0 || a.taps[0].wtap != 0) return -1;
  if ((long)a.N * a.Ho * a.Wo * (long)(a.Ck > a.Cout ? a.Ck : a.Cout) * 2 >= (1L << 32)) return -1;  // 32-bit offsets into the tensors
  for (int i = 0; i < NWG1; ++i)
    if (g_wg1[i].C == a.Ck && g_wg1[i].CO == a.Cout) return i;
  return -1;
}

// 3: the BN-backward sums under a leaky-ReLU mask of slope 0.01 (asm/dconv_gen.py Cfg.stats, LEAKY_BITS)
int wanted_stats(const IgemmArgs& a) { return a.stat_partial == nullptr ? 0 : (a.bn_y != nullptr ? (a.bn_slope != 0.f ? 3 : 2) : 1); }

// the long-reduction pointwise kernel that can run this launch (1x1, stride 1, no addend), or -1
int find_pk(const IgemmArgs& a, int nclass, int stats) {
  if (nclass != 1 || a.IS != 1 || a.OS != 1 || a.pair_delta != 0 || a.wtaps != 1 || a.cls[0].ntaps != 1) return -1;
  if (a.cls[0].taps[0].dh != 0 || a.cls[0].taps[0].dw != 0 || a.cls[0].taps[0].wtap != 0 || a.cls[0].ph != 0 || a.cls[0].pw != 0) return -1;
  if (a.Hsub != a.Hin || a.Wsub != a.Win || a.Hout != a.Hin || a.Wout != a.Win) return -1;
  if (a.pix_stride != a.Ck || a.addend != nullptr || a.q_scale_in != nullptr || a.q_scale_wt != nullptr) return -1;
  const long M = (long)a.N * a.Hin * a.Win;
  for (int i = 0; i < NPK; ++i) {
    const PkVariant& v = g_pk[i];
    if (v.K == a.Ck && v.N == a.Ncols && v.stats == stats && M % v.W == 0 && M / v.W < (1 << 20)) return i;
  }
  return -1;
}

}  // namespace

// MI355_DCONV=0 keeps every launch on the implicit-GEMM kernels (A/B; a switch of struct Knobs: read once and on mi355_reload_knobs()).
// A launch that FORCES an implicit-GEMM tile (MI355_IGEMM8 / MI355_IGEMM_BIG, the per-launch knobs of the tile tests) is left to those
// kernels too, and so is every launch on a device whose runtime refused the embedded code object (dev_state()).
static bool dconv_enabled() {
  return knobs().dconv && !knobs().has_igemm8 && !knobs().has_igemm_big;
}

// plans are also made on GPU-less hosts (layout-only contexts): there the kernels count as available
static bool module_usable() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return true;
  }
  return module_ok();
}

// split count of the generated weight-gradient kernel for this launch (0: not served).  One workgroup per CU: the (ci tile, co tile)
// pairs times the splits fill the device once; a split is a run of whole tiles.
int wg3_plan(int dtype, const WgradArgs& a) {
  if (!knobs().wg3 || !dconv_enabled()) return 0;
  const int vi = find_wg(dtype, a);
  const int v1 = vi < 0 ? find_wg1(dtype, a) : -1;
  if (vi < 0 && v1 < 0) return 0;
  if (!module_usable()) return 0;
  int pairs, ntiles;
  if (vi >= 0) {
    const WgVariant& v = g_wg[vi];
    pairs = (v.C / 64) * (v.CO / 64);
    ntiles = a.N * v.tn / v.ti;
  } else {
    pairs = (g_wg1[v1].C / (64 * g_wg1[v1].XP)) * (g_wg1[v1].CO / (64 * g_wg1[v1].DP));
    ntiles = (a.N * a.Ho * a.Wo + 63) / 64;
  }
  const int cus = device_cus();
  const int max_splits = cus / pairs > 0 ? cus / pairs : 1;
  const int tps = (ntiles + max_splits - 1) / max_splits;
  return (ntiles + tps - 1) / tps;
}

int launch_wg3(const WgradArgs& a, int splits, hipStream_t stream) {
  const int vi = find_wg(MI355_BF16, a);
  const int v1 = vi < 0 ? find_wg1(MI355_BF16, a) : -1;
  MI355_ARG((vi >= 0 || v1 >= 0) && splits >= 1, "wg3: no kernel variant for this launch");
  DevState* d = nullptr;
  if (!dev_state(&d)) return MI355_E_HIP;
  struct __attribute__((packed)) KArgs {
    const void* dy;
    const void* x;
    float* partial;
    unsigned tps, ntiles, npix;
    unsigned pad[7];
  } k;
  if (v1 >= 0) {  // 1x1: flat pixel tiles
    const Wg1Variant& v = g_wg1[v1];
    MI355_ARG((int)sizeof(KArgs) == v.kernarg, "wg1: kernarg size mismatch");
    memset(&k, 0, sizeof(k));
    k.dy = a.dy;
    k.x = a.x;
    k.partial = a.partial;
    k.npix = (unsigned)(a.N * a.Ho * a.Wo);
    k.ntiles = (k.npix + 63) / 64;
    k.tps = (k.ntiles + (unsigned)splits - 1) / (unsigned)splits;
    MI355_ARG((k.ntiles + k.tps - 1) / k.tps == (unsigned)splits, "wg1: %d splits leave an empty split (%u tiles)", splits, k.ntiles);
    size_t ksize = sizeof(k);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
    const hipError_t e = hipModuleLaunchKernel(d->wg1[v1], (unsigned)splits, (unsigned)((v.C / (64 * v.XP)) * (v.CO / (64 * v.DP))), 1, 256, 1, 1, 0, stream, nullptr, extra);
    if (e != hipSuccess) {
      set_error("wg1: hipModuleLaunchKernel(%s) -> %s", v.name, hipGetErrorString(e));
      return MI355_E_HIP;
    }
    note_kernel("%s", v.name);
    return 0;
  }
  const WgVariant& v = g_wg[vi];
  static_assert(sizeof(KArgs) == 64, "kernarg layout of asm/wg_gen.py (Gen.KA)");
  MI355_ARG((int)sizeof(KArgs) == v.kernarg, "wg3: kernarg size mismatch");
  memset(&k, 0, sizeof(k));
  k.dy = a.dy;
  k.x = a.x;
  k.partial = a.partial;
  k.ntiles = (unsigned)(a.N * v.tn / v.ti);
  k.tps = (k.ntiles + (unsigned)splits - 1) / (unsigned)splits;
  MI355_ARG((k.ntiles + k.tps - 1) / k.tps == (unsigned)splits, "wg3: %d splits leave an empty split (%u tiles)", splits, k.ntiles);
  size_t ksize = sizeof(k);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(d->wg[vi], (unsigned)splits, (unsigned)((v.C / 64) * (v.CO / 64)), 1, 256, 1, 1, 0, stream, nullptr, extra);
  if (e != hipSuccess) {
    set_error("wg3: hipModuleLaunchKernel(%s) -> %s", v.name, hipGetErrorString(e));
    return MI355_E_HIP;
  }
  note_kernel("%s", v.name);
  return 0;
}

bool pw_legal(const IgemmArgs& a, int nclass) {
  if (!knobs().pw || !dconv_enabled()) return false;
  return find_pw(a, nclass, wanted_stats(a)) >= 0 && module_ok();
}

int launch_pw(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows) {
  const int vi = find_pw(a, nclass, wanted_stats(a));
  MI355_ARG(vi >= 0, "pw: no kernel variant for this launch");
  const PwVariant& v = g_pw[vi];
  DevState* d = nullptr;
  if (!dev_state(&d)) return MI355_E_HIP;
  struct __attribute__((packed)) KArgs {
    const void* in;
    const void* wt;
    void* out;
    float* stat;
    const void* rsvd[6];
    unsigned units, upw, mtiles, pad0;
    unsigned pad[8];
    unsigned table[128];
  } k;
  static_assert(sizeof(KArgs) == 640, "kernarg layout of asm/pw_gen.py (Gen.KA)");
  MI355_ARG((int)sizeof(KArgs) == v.kernarg, "pw: kernarg size mismatch");
  memset(&k, 0, sizeof(k));
  k.in = a.in;
  k.wt = a.wt;
  k.out = a.out;
  k.stat = a.stat_partial;
  const long M = (long)a.N * a.Hin * a.Win;
  k.mtiles = (unsigned)(M / v.rows);
  k.units = k.mtiles * (unsigned)(v.N / 256);
  const unsigned cus = (unsigned)device_cus();
  k.upw = (k.units + cus - 1) / cus;
  const unsigned grid = (k.units + k.upw - 1) / k.upw;  // every workgroup has at least one unit (it writes its statistics row)
  memcpy(k.table, v.table, sizeof(k.table));
  size_t ksize = sizeof(k);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(d->pw[vi], grid, 1, 1, 256, 1, 1, 0, stream, nullptr, extra);
  if (e != hipSuccess) {
    set_error("pw: hipModuleLaunchKernel(%s) -> %s", v.name, hipGetErrorString(e));
    return MI355_E_HIP;
  }
  if (stat_rows) *stat_rows = a.stat_partial ? (int)grid : 0;
  note_kernel("%s", v.name);
  return 0;
}

bool pk_legal(const IgemmArgs& a, int nclass) {
  if (!knobs().pk || !dconv_enabled()) return false;
  const int v = find_pk(a, nclass, wanted_stats(a));
  if (v < 0) return false;
  const long tiles = (long)a.N * a.Hin * a.Win / g_pk[v].W;   // one partial statistics row per pixel tile
  return (a.stat_partial == nullptr || tiles <= (a.stat_rows_cap > 0 ? a.stat_rows_cap : 768)) && module_ok();
}

int launch_pk(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows) {
  const int vi = find_pk(a, nclass, wanted_stats(a));
  MI355_ARG(vi >= 0, "pk: no kernel variant for this launch");
  const PkVariant& v = g_pk[vi];
  DevState* d = nullptr;
  if (!dev_state(&d)) return MI355_E_HIP;
  struct __attribute__((packed)) KArgs {
    const void* in;
    const void* wt;
    void* out;
    float* stat;
    const void* bn_y;
    const void* bn_bits;
    const float* bn_mean;
    const float* bn_invstd;
    const void* rsvd;
    unsigned nchunks;
    unsigned pad[13];
  } k;
  static_assert(sizeof(KArgs) == 128, "kernarg layout of asm/pk_gen.py (Gen.KA)");
  MI355_ARG((int)sizeof(KArgs) == v.kernarg, "pk: kernarg size mismatch");
  memset(&k, 0, sizeof(k));
  k.in = a.in;
  k.wt = a.wt;
  k.out = a.out;
  k.stat = a.stat_partial;
  k.bn_y = a.bn_y;
  k.bn_bits = a.bn_bits;
  k.bn_mean = a.bn_mean;
  k.bn_invstd = a.bn_invstd;
  k.nchunks = (unsigned)(a.Ck / 64);
  const unsigned tiles = (unsigned)((long)a.N * a.Hin * a.Win / v.W);
  size_t ksize = sizeof(k);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(d->pk[vi], tiles, (unsigned)(v.N / v.BN), 1, 256, 1, 1, 0, stream, nullptr, extra);
  if (e != hipSuccess) {
    set_error("pk: hipModuleLaunchKernel(%s) -> %s", v.name, hipGetErrorString(e));
    return MI355_E_HIP;
  }
  if (stat_rows) *stat_rows = a.stat_partial ? (int)tiles : 0;
  note_kernel("%s", v.name);
  return 0;
}

static bool dconv_legal_of(const IgemmArgs& a, int nclass, int fp8) {
  if (!dconv_enabled()) return false;
  const int v = find_variant(a, nclass, wanted_stats(a), fp8);
  if (v < 0) return false;
  if (fp8 && !knobs().dconv_fp8) return false;
  if (g_variants[v].s2d && !knobs().dconv_s2) return false;
  if (g_variants[v].bnin && !knobs().dconv_bn) return false;
  // one partial statistics row per tile (and class): the caller's buffer must hold them (bn_finalize adds any number of rows, 512 per pass)
  if (a.stat_partial != nullptr && a.N * g_variants[v].TPI / g_variants[v].IPT * (g_variants[v].s2d ? 4 : 1) > (a.stat_rows_cap > 0 ? a.stat_rows_cap : 768)) return false;
  return module_ok();
}

bool dconv_legal(const IgemmArgs& a, int nclass) { return dconv_legal_of(a, nclass, 0); }
// the same launch on e4m3 operands (a.in / a.wt one byte per element; launch_igemm_fp8)
bool dconv_fp8_legal(const IgemmArgs& a, int nclass) { return dconv_legal_of(a, nclass, 1); }

static int launch_dconv_of(const IgemmArgs& a, int nclass, int fp8, float oscale, hipStream_t stream, int* stat_rows);
int launch_dconv(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows) { return launch_dconv_of(a, nclass, 0, 1.f, stream, stat_rows); }
int launch_dconv_fp8(const IgemmArgs& a, int nclass, float oscale, hipStream_t stream, int* stat_rows) { return launch_dconv_of(a, nclass, 1, oscale, stream, stat_rows); }

static int launch_dconv_of(const IgemmArgs& a, int nclass, int fp8, float oscale, hipStream_t stream, int* stat_rows) {
  const int vi = find_variant(a, nclass, wanted_stats(a), fp8);
  MI355_ARG(vi >= 0, "dconv: no kernel variant for this launch");
  const DconvVariant& v = g_variants[vi];
  DevState* d = nullptr;
  if (!dev_state(&d)) return MI355_E_HIP;
  int wtap[9];
  if (v.s2d) s2d_tap_table(a, nclass, wtap);
  else tap_table(a.cls[0], wtap);
  struct __attribute__((packed)) KArgs {
    const void* in;
    const void* wt;
    void* out;
    float* stat;
    const void* bn_y;
    const void* bn_bits;
    const float* bn_mean;
    const float* bn_invstd;
    const void* rsvd;
    unsigned wtap_off[9];
    unsigned nchunks;
    unsigned pad[4];
    unsigned table[768];   // [tile class][wave][64] LDS-DMA piece tables
    unsigned ttable[768];  // bnin kernels only: the transform tables
  } k;
  static_assert(sizeof(KArgs) == 128 + 3072 + 3072, "kernarg layout of asm/dconv_gen.py (Gen.KA; + ttables for Cfg.bnin)");
  size_t ksize = v.bnin ? sizeof(KArgs) : sizeof(KArgs) - sizeof(k.ttable);
  MI355_ARG((int)ksize == v.kernarg, "dconv: kernarg size mismatch");
  k.in = a.in;
  k.wt = a.wt;
  k.out = a.out;
  k.stat = a.stat_partial;
  k.bn_y = a.bn_y;
  k.bn_bits = a.bn_bits;
  k.bn_mean = a.bn_mean;
  k.bn_invstd = a.bn_invstd;
  k.rsvd = nullptr;
  memset(k.pad, 0, sizeof(k.pad));
  memcpy(k.table, v.table, sizeof(k.table));
  if (v.bnin) {  // the pointer slots of the BN-backward sums carry the input's BatchNorm: a out, its bits out, [2][Ck] scale / shift
    k.bn_y = a.bn_in_a;
    k.bn_bits = a.bn_in_bits;
    k.bn_mean = a.bn_in;
    k.bn_invstd = nullptr;
    int ti = -1;
    for (int i = 0; i < NTT; ++i)
      if (strcmp(g_tt[i].name, v.name) == 0) ti = i;
    MI355_ARG(ti >= 0, "dconv: no transform table for %s", v.name);
    memcpy(k.ttable, g_tt[ti].table, sizeof(k.ttable));
  }
  const int es = v.fp8 ? 1 : 2;
  for (int t = 0; t < 9; ++t) k.wtap_off[t] = (unsigned)(wtap[t] * a.Ck * es);
  k.nchunks = (unsigned)(a.Ck * es / 128);   // chunks of one 128-byte LDS row: 64 bf16 / 128 e4m3 channels
  if (v.fp8) {  // out = accumulator * oscale / (*q_scale_in * *q_scale_wt) (both null: oscale alone); kernarg slots rsvd, pad[1], pad[2:3]
    k.rsvd = a.q_scale_in;
    memcpy(&k.pad[1], &oscale, sizeof(float));
    memcpy(&k.pad[2], &a.q_scale_wt, sizeof(void*));
  }
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
  const int tiles = a.N * v.TPI / v.IPT;
  const int ncls = v.s2d ? 4 : 1;   // workgroup id y = class * column tiles + column tile: the long classes (4 taps) are dispatched first
  const hipError_t e = hipModuleLaunchKernel(d->fn[vi], (unsigned)tiles, (unsigned)(a.Ncols / v.BN * ncls), 1, 256, 1, 1, 0, stream, nullptr, extra);
  if (e != hipSuccess) {
    set_error("dconv: hipModuleLaunchKernel(%s) -> %s", v.name, hipGetErrorString(e));
    return MI355_E_HIP;
  }
  if (stat_rows) *stat_rows = a.stat_partial ? tiles * ncls : 0;
  note_kernel("%s", v.name);
  return 0;
}

// ---- output-heavy pointwise kernels with resident weights (asm/po_gen.py) ---------------------------------------------------------
namespace {

struct PoPlan {
  int vi = -1;
  unsigned T = 0, nct = 0, tpg = 0, G = 0, grid = 0, lognct = 0;
};

// 1x1 / stride 1, K = the variant's, columns a power-of-two multiple of its BN; statistics 0 / 1 / 2; addend none / plain / under its mask
bool plan_po(const IgemmArgs& a, int nclass, PoPlan* pl) {
  if (nclass != 1 || a.IS != 1 || a.OS != 1 || a.pair_delta != 0 || a.wtaps != 1 || a.cls[0].ntaps != 1) return false;
  if (a.cls[0].taps[0].dh != 0 || a.cls[0].taps[0].dw != 0 || a.cls[0].taps[0].wtap != 0 || a.cls[0].ph != 0 || a.cls[0].pw != 0) return false;
  if (a.Hsub != a.Hin || a.Wsub != a.Win || a.Hout != a.Hin || a.Wout != a.Win) return false;
  if (a.pix_stride != a.Ck || a.q_scale_in != nullptr || a.q_scale_wt != nullptr) return false;
  const int stats = wanted_stats(a);
  const int add = a.addend == nullptr ? 0 : (a.addend_sub2 ? 3 : (a.addend_bits != nullptr ? 2 : 1));
  if (a.addend == nullptr && (a.addend_bits != nullptr || a.addend_sub2)) return false;
  if (a.addend_sub2 && (a.addend_bits != nullptr || a.Hout % 2 || a.Wout % 2)) return false;
  const long M = (long)a.N * a.Hin * a.Win;
  if (M * a.Ck * 2 >= (1L << 32) || M * a.Ncols * 2 >= (1L << 32)) return false;  // 32-bit num_records of the tile-by-tile descriptors
  const int bnin = a.bn_in != nullptr ? 1 : 0;
  if (bnin && (a.bn_in_a == nullptr || a.bn_in_bits == nullptr)) return false;
  if (a.addend_sub2 && (M + 64) * (a.Wout > a.Hout ? a.Wout : a.Hout) >= (1L << 32)) return false;  // exactness of the kernel's divisions
  for (int i = 0; i < NPO; ++i) {
    const PoVariant& v = g_po[i];
    if (v.K != a.Ck || v.stats != stats || v.add != add || a.Ncols % v.BN != 0 || v.bnin != bnin) continue;
    if (v.bnin && M % v.TP != 0) continue;   // (a ragged tile's missing pixels would become relu(shift) instead of zero)
    const unsigned nct = (unsigned)(a.Ncols / v.BN);
    if ((nct & (nct - 1)) != 0 || nct > 32) continue;
    pl->vi = i;
    pl->nct = nct;
    pl->lognct = 0;
    while ((1u << pl->lognct) < nct) ++pl->lognct;
    pl->T = (unsigned)((M + v.TP - 1) / v.TP);
    const unsigned cus = (unsigned)device_cus();
    const unsigned gmax = cus / nct > 0 ? cus / nct : 1;       // pixel runs: one workgroup per CU over all column tiles
    pl->tpg = (pl->T + gmax - 1) / gmax;
    pl->G = (pl->T + pl->tpg - 1) / pl->tpg;
    pl->grid = (pl->G + 7) / 8 * 8 * nct;                      // workgroup x: XCD x % 8, column tile (x / 8) % nct, run (x / 8 / nct) * 8 + x % 8
    return true;
  }
  return false;
}

}  // namespace

bool po_legal(const IgemmArgs& a, int nclass) {
  const int mode = knobs().po;
  if (!mode || !dconv_enabled()) return false;
  PoPlan pl;
  if (!plan_po(a, nclass, &pl)) return false;
  if (a.stat_partial != nullptr && (int)pl.G * g_po[pl.vi].WM > (a.stat_rows_cap > 0 ? a.stat_rows_cap : 768)) return false;
  if (g_po[pl.vi].bnin && !knobs().po_bn) return false;
  if (mode < 2) {
    // MI355_PO=1 (default): the measured rule, per launch shape of the bs-256 step in a serial trace (profiles/r05_ab_po_*.txt):
    //  - 64 -> 256 under the shortcut addend + BN-backward sums (layer 1's conv1 data gradient, 1.39 GB per launch): the implicit-GEMM
    //    kernel streams it at 5.5 TB/s with two workgroups per CU out of phase, this kernel at 5.1: stays there;
    //  - (K = 512 with an addend, layer 4's conv1 data gradient: 74 us against 67 with one mask byte load per lane and item; 63 with the
    //    tile-wide mask loads: served here since)
    //  - 512 -> 2048 forward (layer 4's conv3): pk's four-image tiles win by 2 us.
    const PoVariant& v = g_po[pl.vi];
    if (v.BN == 64 && knobs().po64 < 2) {
      if (!knobs().po64) return false;
    }
    // 64 -> 256 under the shortcut addend + sums (layer 1's conv1 data gradient): a tie per launch since the tile-wide mask loads (250-252 us here,
    // 250-257 on the implicit-GEMM kernel), but the step is 0.06 ms SLOWER with it here (18.03 -> 18.09, four alternations on one box): the 768-workgroup
    // grid shares the chip with the weight-gradient stream better than 256 persistent workgroups do
    if (v.K == 64 && v.add != 0 && v.stats == 2) return false;
    if (v.K == 512 && a.Ncols >= 2048 && pk_legal(a, nclass)) return false;
  }
  return module_ok();
}

bool igemm_sub2_legal(int dtype, const IgemmArgs& a, int nclass) { return dtype == MI355_BF16 && a.addend_sub2 && po_legal(a, nclass); }

// a launch with the input's BatchNorm + ReLU in the operand path (IgemmArgs::bn_in) has a kernel: the direct 3x3 kernels (conv2 <- bn1) or the
// resident-weight pointwise kernels (conv3 <- bn2)
bool igemm_bn_in_legal(int dtype, const IgemmArgs& a, int nclass) {
  return dtype == MI355_BF16 && a.bn_in != nullptr && (dconv_legal(a, nclass) || po_legal(a, nclass));
}

int launch_po(const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows) {
  PoPlan pl;
  MI355_ARG(plan_po(a, nclass, &pl), "po: no kernel variant for this launch");
  const PoVariant& v = g_po[pl.vi];
  DevState* d = nullptr;
  if (!dev_state(&d)) return MI355_E_HIP;
  struct __attribute__((packed)) KArgs {
    const void* in;
    const void* wt;
    void* out;
    float* stat;
    const void* bn_y;
    const void* bn_bits;
    const float* bn_mean;
    const float* bn_invstd;
    const void* addend;
    const void* addend_bits;
    unsigned npix, ncols, tpg, ngroups, ntiles, lognct, W, H, magic_w, magic_h;
    unsigned pad[2];
  } k;
  static_assert(sizeof(KArgs) == 128, "kernarg layout of asm/po_gen.py (Gen.KA)");
  MI355_ARG((int)sizeof(KArgs) == v.kernarg, "po: kernarg size mismatch");
  memset(&k, 0, sizeof(k));
  k.in = a.in;
  k.wt = a.wt;
  k.out = a.out;
  k.stat = a.stat_partial;
  k.bn_y = a.bn_y;
  k.bn_bits = a.bn_bits;
  k.bn_mean = a.bn_mean;
  k.bn_invstd = a.bn_invstd;
  if (v.bnin) {  // the pointer slots of the BN-backward sums carry the input's BatchNorm: a out, its bits out, [2][Ck] scale / shift
    k.bn_y = a.bn_in_a;
    k.bn_bits = a.bn_in_bits;
    k.bn_mean = a.bn_in;
    k.bn_invstd = nullptr;
  }
  k.addend = a.addend;
  k.addend_bits = a.addend_bits;
  k.npix = (unsigned)((long)a.N * a.Hin * a.Win);
  k.ncols = (unsigned)a.Ncols;
  k.tpg = pl.tpg;
  k.ngroups = pl.G;
  k.ntiles = pl.T;
  k.lognct = pl.lognct;
  k.W = (unsigned)a.Wout;
  k.H = (unsigned)a.Hout;
  k.magic_w = (unsigned)((1ull << 32) / (unsigned)a.Wout + 1);  // q / W = mulhi(q, magic) while q * W < 2^32 (plan_po checks)
  k.magic_h = (unsigned)((1ull << 32) / (unsigned)a.Hout + 1);
  size_t ksize = sizeof(k);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksize, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(d->po[pl.vi], pl.grid, 1, 1, 256, 1, 1, 0, stream, nullptr, extra);
  if (e != hipSuccess) {
    set_error("po: hipModuleLaunchKernel(%s) -> %s", v.name, hipGetErrorString(e));
    return MI355_E_HIP;
  }
  if (stat_rows) *stat_rows = a.stat_partial ? (int)pl.G * v.WM : 0;
  note_kernel("%s", v.name);
  return 0;
}

}  // namespace mi355
