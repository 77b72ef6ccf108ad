// conv_api.cpp — geometry builders and the per-op C-ABI entry points declared in include/mi355rn.h.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>

#include "common.h"

namespace mi355 {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
const char* get_error() { return g_err.c_str(); }

int probe_env(const char* name) {
  const char* v = getenv(name);
  const int x = v ? atoi(v) : 0;
  if (x != 0) {
    static std::mutex mu;
    static std::set<std::string> warned;
    std::lock_guard<std::mutex> lock(mu);
    if (warned.insert(name).second)
      fprintf(stderr, "libmi355rn: %s=%d — timing probe active: kernels skip loads / stores, RESULTS ARE NOT VALID\n", name, x);
  }
  return x;
}

// compute units of the current device (hipDeviceProp), the unit every persistent-grid size in this library is a multiple of;
// 256 (MI355X) while no device is visible — layout-only contexts plan on a GPU-less host
// CUs left to a communication library's kernels (mi355_set_reserved_cus / MI355_RESERVE_CUS, a multiple of 8 = the same number per XCD):
// every grid in this library that is sized from the CU count (persistent implicit-GEMM grids, the split plans of the weight-gradient
// kernels, the pointwise kernels' units per workgroup) then plans for CUs - reserved.  Grids that are tile counts (one image per
// workgroup) do not change.  Process-global; set before contexts are created (split plans are made at context creation).
static std::atomic<int> g_reserved_cus{-1};

int device_cus() {
  static int cached[64] = {0};
  int n = 0, dev = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 256;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cached[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cached[dev] = v;
  }
  int r = g_reserved_cus.load();
  if (r < 0) {  // first use: the environment's value (A/B runs), else none
    const char* e = getenv("MI355_RESERVE_CUS");
    r = e ? atoi(e) / 8 * 8 : 0;
    if (r < 0) r = 0;
    g_reserved_cus.store(r);
  }
  const int cus = cached[dev] ? cached[dev] : 256;
  return cus - r >= 64 ? cus - r : (cus >= 64 ? 64 : cus);
}

extern "C" int mi355_set_reserved_cus(int n) {
  MI355_ARG(n >= 0 && n % 8 == 0, "set_reserved_cus: %d (a non-negative multiple of 8: the same number per XCD)", n);
  g_reserved_cus.store(n);
  return 0;
}

static Knobs g_knobs;
static std::atomic<bool> g_knobs_loaded{false};
static std::mutex g_knobs_mu;

static void load_knobs() {
  Knobs k;
  if (const char* e = getenv("MI355_IGEMM8")) {
    k.has_igemm8 = true;
    strncpy(k.igemm8, e, sizeof(k.igemm8) - 1);
  }
  if (const char* e = getenv("MI355_IGEMM_BIG")) {
    k.has_igemm_big = true;
    k.igemm_big = atoi(e);
  }
  if (const char* e = getenv("MI355_STEM_TH")) k.stem_th = atoi(e);
  k.stem_dbg = getenv("MI355_STEM_DBG") != nullptr;
  if (const char* e = getenv("MI355_SK_DEBUG")) {
    if (strcmp(e, "mute") == 0) k.sk_mute = true;
    else if (!k.error[0]) snprintf(k.error, sizeof(k.error), "MI355_SK_DEBUG=%s: not \"mute\"", e);
  }
  // switches with a closed domain: anything else is an error, not a silent default (a stale MI355_DCONV=off in a job script would
  // otherwise cost a millisecond per step without a word)
  auto sw = [&](const char* name, int* dst, int maxv) {
    const char* e = getenv(name);
    if (!e) return;
    if (e[0] >= '0' && e[0] <= '0' + maxv && e[1] == 0) {
      *dst = e[0] - '0';
    } else if (!k.error[0]) {
      snprintf(k.error, sizeof(k.error), "%s=%s: not one of 0..%d", name, e, maxv);
    }
  };
  sw("MI355_STEM_DIRECT", &k.stem_direct, 1);
  sw("MI355_DCONV", &k.dconv, 1);
  sw("MI355_WG3", &k.wg3, 1);
  sw("MI355_PW", &k.pw, 1);
  sw("MI355_PK", &k.pk, 1);
  sw("MI355_PO", &k.po, 2);
  sw("MI355_PO64", &k.po64, 2);
  sw("MI355_DCONV_S2", &k.dconv_s2, 1);
  sw("MI355_DCONV_BN", &k.dconv_bn, 1);
  sw("MI355_PO_BN", &k.po_bn, 1);
  sw("MI355_POOL_KEYS", &k.pool_keys, 1);
  sw("MI355_DCONV_FP8", &k.dconv_fp8, 1);
  if (k.has_igemm_big && k.igemm_big != 0 && k.igemm_big != 1 && k.igemm_big != 3 && !k.error[0])
    snprintf(k.error, sizeof(k.error), "MI355_IGEMM_BIG=%d: not one of 0, 1, 3", k.igemm_big);
  g_knobs = k;
}

const Knobs& knobs() {
  if (!g_knobs_loaded.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_knobs_mu);
    if (!g_knobs_loaded.load()) {
      load_knobs();
      g_knobs_loaded.store(true, std::memory_order_release);
    }
  }
  return g_knobs;
}

// re-reads the per-launch tile knobs (tests flip them between launches of one process; not thread-safe against running launches)
extern "C" int mi355_reload_knobs(void) {
  std::lock_guard<std::mutex> lock(g_knobs_mu);
  load_knobs();
  g_knobs_loaded.store(true, std::memory_order_release);
  MI355_ARG(!g_knobs.error[0], "%s", g_knobs.error);
  return 0;
}

static thread_local char g_last_kernel[96] = "";
void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
  va_end(ap);
  static const bool trace = getenv("MI355_TRACE_KERNELS") != nullptr;  // debugging aid: one line per conv / weight-gradient launch
  if (trace) fprintf(stderr, "mi355 kernel: %s\n", g_last_kernel);
  static const bool dsync = getenv("MI355_TRACE_SYNC") != nullptr;  // debugging aid: every conv launch drains the device (exposes missing stream dependencies)
  if (dsync) (void)hipDeviceSynchronize();
}
extern "C" const char* mi355_last_conv_kernel(void) { return g_last_kernel; }

static int out_dim(int H, int K, int s, int p) { return (H + 2 * p - K) / s + 1; }

void build_fwd_args(IgemmArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  memset(&a, 0, sizeof(a));
  const int Ho = out_dim(H, KH, stride, pad), Wo = out_dim(W, KW, stride, pad);
  a.N = N; a.Hin = H; a.Win = W; a.pix_stride = Cin;
  a.Hsub = Ho; a.Wsub = Wo; a.IS = stride;
  a.Hout = Ho; a.Wout = Wo; a.OS = 1;
  a.Ck = Cin; a.Ncols = Cout; a.wtaps = KH * KW;
  TapClass& c = a.cls[0];
  c.ph = c.pw = 0;
  c.ntaps = KH * KW;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      Tap& t = c.taps[kh * KW + kw];
      t.dh = (int8_t)(kh - pad);
      t.dw = (int8_t)(kw - pad);
      t.wtap = (int16_t)(kh * KW + kw);
    }
}

// returns the number of tap classes (1 for stride 1, 4 for stride 2), or a negative status
int build_dgrad_args(IgemmArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  memset(&a, 0, sizeof(a));
  const int Ho = out_dim(H, KH, stride, pad), Wo = out_dim(W, KW, stride, pad);
  a.N = N; a.Hin = Ho; a.Win = Wo; a.pix_stride = Cout;
  a.IS = 1; a.Hout = H; a.Wout = W;
  a.Ck = Cout; a.Ncols = Cin; a.wtaps = KH * KW;
  if (stride == 1) {
    a.OS = 1; a.Hsub = H; a.Wsub = W;
    TapClass& c = a.cls[0];
    c.ph = c.pw = 0;
    c.ntaps = KH * KW;
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) {
        Tap& t = c.taps[kh * KW + kw];
        t.dh = (int8_t)(pad - kh);
        t.dw = (int8_t)(pad - kw);
        t.wtap = (int16_t)(kh * KW + kw);
      }
    return 1;
  }
  MI355_ARG(stride == 2 && H % 2 == 0 && W % 2 == 0, "dgrad: stride=%d with H=%d W=%d unsupported", stride, H, W);
  a.OS = 2; a.Hsub = H / 2; a.Wsub = W / 2;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      TapClass& c = a.cls[ph * 2 + pw];
      c.ph = ph; c.pw = pw; c.ntaps = 0;
      for (int kh = 0; kh < KH; ++kh) {
        const int vh = ph + pad - kh;
        if (vh & 1) continue;
        for (int kw = 0; kw < KW; ++kw) {
          const int vw = pw + pad - kw;
          if (vw & 1) continue;
          Tap& t = c.taps[c.ntaps++];
          t.dh = (int8_t)(vh / 2);
          t.dw = (int8_t)(vw / 2);
          t.wtap = (int16_t)(kh * KW + kw);
        }
      }
    }
  // longest classes first: the persistent workgroups take items in class order, so the last (partial) round is then
  // made of the short items (3x3: 4, 2, 2, 1 taps) instead of the long ones
  for (int i = 0; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j)
      if (a.cls[j].ntaps > a.cls[i].ntaps) {
        const TapClass t = a.cls[i];
        a.cls[i] = a.cls[j];
        a.cls[j] = t;
      }
  return 4;
}

void build_wgrad_args(WgradArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  memset(&a, 0, sizeof(a));
  a.N = N; a.Ho = out_dim(H, KH, stride, pad); a.Wo = out_dim(W, KW, stride, pad); a.Cout = Cout;
  a.Hin = H; a.Win = W; a.pix_stride = Cin; a.IS = stride;
  a.Ck = Cin; a.wtaps = KH * KW; a.ntaps = KH * KW;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      Tap& t = a.taps[kh * KW + kw];
      t.dh = (int8_t)(kh - pad);
      t.dw = (int8_t)(kw - pad);
      t.wtap = (int16_t)(kh * KW + kw);
    }
}

void build_stem_fwd_args(IgemmArgs& a, int N, int H, int W) {
  memset(&a, 0, sizeof(a));
  a.N = N; a.Hin = stem_hp(H); a.Win = stem_wp(W); a.pix_stride = STEM_PS;
  a.Hsub = H / 2; a.Wsub = W / 2; a.IS = 2;
  a.Hout = H / 2; a.Wout = W / 2; a.OS = 1;
  // a tap = TWO image rows: k in [0,32) is (kw*4+c) of row 2t, k in [32,64) the same of row 2t+1 (kh = 7 has zero
  // weights; the padded image has the row)
  a.Ck = STEM_CK; a.Ncols = 64; a.wtaps = 4;
  a.pair_delta = stem_wp(W) * STEM_PS - 32;
  TapClass& c = a.cls[0];
  c.ntaps = 4;
  for (int t = 0; t < 4; ++t) {
    c.taps[t].dh = (int8_t)(2 * t);
    c.taps[t].dw = 0;
    c.taps[t].wtap = (int16_t)t;
  }
}

void build_stem_wgrad_args(WgradArgs& a, int N, int H, int W) {
  memset(&a, 0, sizeof(a));
  a.N = N; a.Ho = H / 2; a.Wo = W / 2; a.Cout = 64;
  a.Hin = stem_hp(H); a.Win = stem_wp(W); a.pix_stride = STEM_PS; a.IS = 2;
  a.Ck = STEM_CK; a.wtaps = 4; a.ntaps = 4;
  a.pair_delta = stem_wp(W) * STEM_PS - 32;
  for (int t = 0; t < 4; ++t) {
    a.taps[t].dh = (int8_t)(2 * t);
    a.taps[t].dw = 0;
    a.taps[t].wtap = (int16_t)t;
  }
}

static int check_conv(int dtype, int Cin, int Cout, int KH, int KW, int stride) {
  MI355_ARG(dtype == MI355_F32 || dtype == MI355_BF16, "conv: bad dtype %d", dtype);
  MI355_ARG(Cin % 64 == 0 && Cout % 64 == 0, "conv: Cin=%d Cout=%d must be multiples of 64", Cin, Cout);
  MI355_ARG(KH * KW <= 9 && KH >= 1 && KW >= 1, "conv: kernel %dx%d unsupported (<= 9 taps)", KH, KW);
  MI355_ARG(stride == 1 || stride == 2, "conv: stride %d unsupported", stride);
  return 0;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

const char* mi355_last_error(void) { return mi355::get_error(); }
int mi355_version(void) { return 100; }
int mi355_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

size_t mi355_conv2d_workspace_bytes(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                    int pad) {
  const size_t es = dtype_size(dtype);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const size_t wbytes = align_up((size_t)Cin * KH * KW * Cout * es, 256);
  WgradArgs wa;
  build_wgrad_args(wa, N, H, W, Cin, Cout, KH, KW, stride, pad);
  const int splits = std::max(plan_wgrad(dtype, wa), plan_wgrad_splits(dtype, N * Ho * Wo, Cout, KH * KW, Cin));  // (fp8 twins use the latter)
  const size_t pbytes = align_up((size_t)splits * Cout * KH * KW * Cin * 4, 256);
  return wbytes + pbytes + wbytes;  // transposed + partials + cast copy
}

int mi355_conv2d_fwd(int dtype, const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int KH,
                     int KW, int stride, int pad, void* stream) {
  MI355_TRY(check_conv(dtype, Cin, Cout, KH, KW, stride));
  IgemmArgs a;
  build_fwd_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  a.in = x; a.wt = w; a.out = y;
  return launch_igemm(dtype, a, 1, (hipStream_t)stream);
}

int mi355_conv2d_fwd_stats(int dtype, const void* x, const void* w, void* y, float* partial, size_t partial_bytes, int* nblk, int N,
                           int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  MI355_TRY(check_conv(dtype, Cin, Cout, KH, KW, stride));
  MI355_ARG(partial && nblk && partial_bytes >= (size_t)768 * 2 * Cout * sizeof(float), "conv2d_fwd_stats: partial buffer needs 768 * 2 * Cout floats");
  IgemmArgs a;
  build_fwd_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  a.in = x; a.wt = w; a.out = y;
  a.stat_partial = partial;
  a.stat_rows_cap = (int)std::min<size_t>(partial_bytes / ((size_t)2 * Cout * sizeof(float)), 1u << 20);  // (the generated kernels write one row per pixel tile)
  return launch_igemm(dtype, a, 1, (hipStream_t)stream, nblk);
}

int mi355_conv2d_fwd_bn_in(int dtype, const void* y_in, const float* scale_shift, const void* w, void* y, void* a_out, uint8_t* a_bits, float* partial,
                           size_t partial_bytes, int* nblk, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  MI355_TRY(check_conv(dtype, Cin, Cout, KH, KW, stride));
  MI355_ARG(y_in && scale_shift && a_out && a_bits, "conv2d_fwd_bn_in: null pointer");
  MI355_ARG(partial && nblk && partial_bytes >= (size_t)768 * 2 * Cout * sizeof(float), "conv2d_fwd_bn_in: partial buffer needs 768 * 2 * Cout floats");
  IgemmArgs a;
  build_fwd_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  a.in = y_in; a.wt = w; a.out = y;
  a.bn_in = scale_shift; a.bn_in_a = a_out; a.bn_in_bits = a_bits;
  a.stat_partial = partial;
  a.stat_rows_cap = (int)std::min<size_t>(partial_bytes / ((size_t)2 * Cout * sizeof(float)), 1u << 20);
  MI355_ARG(igemm_bn_in_legal(dtype, a, 1), "conv2d_fwd_bn_in: no kernel applies the input's BatchNorm for this launch (bf16; 3x3 / stride 1 at a generated shape, or 1x1 with 64 / 128 / 256 input channels into a multiple of 256 columns over whole 64-pixel tiles)");
  return launch_igemm(dtype, a, 1, (hipStream_t)stream, nblk);
}

int mi355_conv2d_dgrad(int dtype, const void* dy, const void* w, void* dx, const void* addend, int N, int H, int W,
                       int Cin, int Cout, int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes,
                       void* stream) {
  MI355_TRY(check_conv(dtype, Cin, Cout, KH, KW, stride));
  const size_t es = dtype_size(dtype);
  const size_t wbytes = (size_t)Cin * KH * KW * Cout * es;
  MI355_ARG(ws && ws_bytes >= wbytes, "dgrad: workspace too small (%zu < %zu)", ws_bytes, wbytes);
  hipStream_t s = (hipStream_t)stream;
  // transpose [Cout][taps][Cin] -> [Cin][taps][Cout] in the same dtype
  MI355_TRY(launch_transpose_any(dtype, w, ws, Cout, KH * KW, Cin, s));
  IgemmArgs a;
  const int nclass = build_dgrad_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  if (nclass < 0) return nclass;
  a.in = dy; a.wt = ws; a.out = dx; a.addend = addend;
  return launch_igemm(dtype, a, nclass, s);
}

int mi355_conv2d_dgrad_bn(int dtype, const void* dy, const void* w, void* dx, const void* addend, const uint8_t* addend_bits, int addend_sub2, const void* bn_y,
                          const uint8_t* bn_bits, const float* bn_mean, const float* bn_invstd, float* partial, size_t partial_bytes, int* nblk, int N,
                          int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  return mi355_conv2d_dgrad_bn_leaky(dtype, dy, w, dx, addend, addend_bits, addend_sub2, bn_y, bn_bits, bn_mean, bn_invstd, 0.f, partial, partial_bytes, nblk, N, H, W,
                                     Cin, Cout, KH, KW, stride, pad, ws, ws_bytes, stream);
}

int mi355_conv2d_dgrad_bn_leaky(int dtype, const void* dy, const void* w, void* dx, const void* addend, const uint8_t* addend_bits, int addend_sub2,
                                const void* bn_y, const uint8_t* bn_bits, const float* bn_mean, const float* bn_invstd, float slope, float* partial,
                                size_t partial_bytes, int* nblk, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* ws,
                                size_t ws_bytes, void* stream) {
  MI355_TRY(check_conv(dtype, Cin, Cout, KH, KW, stride));
  MI355_ARG(slope == 0.f || partial, "dgrad_bn_leaky: a slope without the BN-backward sums");
  const size_t es = dtype_size(dtype);
  const size_t wbytes = (size_t)Cin * KH * KW * Cout * es;
  MI355_ARG(ws && ws_bytes >= wbytes, "dgrad_bn: workspace too small (%zu < %zu)", ws_bytes, wbytes);
  MI355_ARG(addend || !addend_bits, "dgrad_bn: addend_bits without an addend");
  MI355_ARG(!partial || (bn_y && bn_bits && bn_mean && bn_invstd && nblk), "dgrad_bn: the BN-backward sums need bn_y, bn_bits, bn_mean, bn_invstd and nblk");
  MI355_ARG(!partial || partial_bytes >= (size_t)768 * 2 * Cin * sizeof(float), "dgrad_bn: partial buffer needs 768 * 2 * Cin floats");
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(launch_transpose_any(dtype, w, ws, Cout, KH * KW, Cin, s));
  IgemmArgs a;
  const int nclass = build_dgrad_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  if (nclass < 0) return nclass;
  a.in = dy; a.wt = ws; a.out = dx; a.addend = addend; a.addend_bits = addend_bits; a.addend_sub2 = addend_sub2;
  MI355_ARG(!addend_sub2 || igemm_sub2_legal(dtype, a, nclass), "dgrad_bn: no kernel takes a half-resolution addend for this launch (bf16 1x1 / stride 1, even H and W)");
  if (partial) {
    a.stat_partial = partial;
    a.stat_rows_cap = (int)std::min<size_t>(partial_bytes / ((size_t)2 * Cin * sizeof(float)), 1u << 20);
    a.bn_y = bn_y; a.bn_bits = bn_bits; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd; a.bn_slope = slope;
  }
  if (nblk) *nblk = 0;
  return launch_igemm(dtype, a, nclass, s, partial ? nblk : nullptr);
}

int mi355_conv2d_wgrad(int dtype, const void* dy, const void* x, float* dw, float beta, int N, int H, int W, int Cin,
                       int Cout, int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  MI355_TRY(check_conv(dtype, Cin, Cout, KH, KW, stride));
  WgradArgs a;
  build_wgrad_args(a, N, H, W, Cin, Cout, KH, KW, stride, pad);
  const int splits = plan_wgrad(dtype, a);
  const size_t n = (size_t)Cout * KH * KW * Cin;
  MI355_ARG(ws && ws_bytes >= (size_t)splits * n * 4, "wgrad: workspace too small (%zu < %zu)", ws_bytes,
            (size_t)splits * n * 4);
  a.dy = dy; a.x = x; a.partial = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(launch_wgrad(dtype, a, splits, s));
  return launch_splitk_reduce((const float*)ws, splits, n, dw, n, beta, s);
}

size_t mi355_stem_xpad_bytes(int dtype, int N, int H, int W) {
  return (size_t)N * stem_hp(H) * stem_wp(W) * STEM_PS * dtype_size(dtype);
}

size_t mi355_stem_workspace_bytes(int dtype, int N, int H, int W) {
  const int splits = plan_wgrad_splits(dtype, N * (H / 2) * (W / 2), 64, 4, STEM_CK);
  return align_up((size_t)64 * 4 * 64 * dtype_size(dtype), 256) + (size_t)splits * 64 * 4 * 64 * 4;
}

int mi355_stem_ingest(int dtype, const float* x_nchw, void* xpad, int N, int H, int W, void* stream) {
  MI355_ARG(x_nchw && xpad && H % 2 == 0 && W % 2 == 0, "stem_ingest: bad arguments");
  return launch_stem_ingest(dtype, x_nchw, xpad, N, H, W, (hipStream_t)stream);
}

int mi355_stem_fwd(int dtype, const void* xpad, const float* w_krsc, void* y, int N, int H, int W, void* ws,
                   size_t ws_bytes, void* stream) {
  const size_t pk = (size_t)64 * 4 * 64 * dtype_size(dtype);
  MI355_ARG(ws && ws_bytes >= pk, "stem_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(launch_stem_pack(dtype, w_krsc, ws, s));
  IgemmArgs a;
  build_stem_fwd_args(a, N, H, W);
  a.in = xpad; a.wt = ws; a.out = y;
  return launch_igemm(dtype, a, 1, s);
}

int mi355_stem_wgrad(int dtype, const void* dy, const void* xpad, float* dw, float beta, int N, int H, int W, void* ws,
                     size_t ws_bytes, void* stream) {
  WgradArgs a;
  build_stem_wgrad_args(a, N, H, W);
  const int splits = plan_wgrad_splits(dtype, a.N * a.Ho * a.Wo, 64, 4, STEM_CK);
  MI355_ARG(ws && ws_bytes >= (size_t)splits * 64 * 4 * 64 * 4, "stem_wgrad: workspace too small");
  a.dy = dy; a.x = xpad; a.partial = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(launch_wgrad(dtype, a, splits, s));
  return launch_stem_unpack((float*)ws, splits, dw, beta, s);
}

size_t mi355_bn_workspace_bytes(int C) { return ((size_t)bn_max_blocks() * 2 * C + 8 * (size_t)C) * 4; }

int mi355_bn_fwd_train(int dtype, const void* x, const void* residual, void* out, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float* save_mean,
                       float* save_invstd, int M, int C, float eps, float momentum, int relu, void* ws,
                       size_t ws_bytes, void* stream) {
  MI355_ARG(ws && ws_bytes >= mi355_bn_workspace_bytes(C), "bn: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)ws;
  float* scale = partial + (size_t)bn_max_blocks() * 2 * C;
  float* shift = scale + C;
  float* pivot = shift + C;
  int nblk = 0;
  MI355_TRY(launch_bn_stats(dtype, x, partial, pivot, &nblk, M, C, s));
  MI355_TRY(launch_bn_finalize(partial, pivot, nblk, M, C, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                               scale, shift, eps, momentum, s));
  return launch_bn_apply(dtype, x, scale, shift, residual, nullptr, nullptr, nullptr, out, M, C, relu, s);
}

int mi355_bn_fwd_train_partial(int dtype, const void* x, const void* residual, void* out, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd, int M, int C, float eps,
                               float momentum, int relu, const float* partial, int nblk, void* ws, size_t ws_bytes, void* stream) {
  MI355_ARG(ws && ws_bytes >= (size_t)2 * C * 4 && partial && nblk >= 1, "bn_fwd_train_partial: nblk=%d, workspace of 2*C floats", nblk);
  hipStream_t s = (hipStream_t)stream;
  float* scale = (float*)ws;
  float* shift = scale + C;
  MI355_TRY(launch_bn_finalize(partial, nullptr, nblk, M, C, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale,
                               shift, eps, momentum, s));
  return launch_bn_apply(dtype, x, scale, shift, residual, nullptr, nullptr, nullptr, out, M, C, relu, s);
}

int mi355_bn_fwd_eval(int dtype, const void* x, const void* residual, void* out, const float* gamma,
                      const float* beta, const float* running_mean, const float* running_var, int M, int C, float eps,
                      int relu, void* ws, size_t ws_bytes, void* stream) {
  MI355_ARG(ws && ws_bytes >= (size_t)2 * C * 4, "bn_eval: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* scale = (float*)ws;
  float* shift = scale + C;
  MI355_TRY(launch_bn_eval_coeffs(gamma, beta, running_mean, running_var, scale, shift, C, eps, s));
  return launch_bn_apply(dtype, x, scale, shift, residual, nullptr, nullptr, nullptr, out, M, C, relu, s);
}

int mi355_bn_bwd(int dtype, const void* dout, const void* out, const void* x, const float* gamma,
                 const float* save_mean, const float* save_invstd, void* dx, void* dz_out, float* dgamma,
                 float* dbeta, float beta_acc, int M, int C, int relu, void* ws, size_t ws_bytes, void* stream) {
  MI355_ARG(ws && ws_bytes >= mi355_bn_workspace_bytes(C), "bn: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)ws;
  float* coef = partial + (size_t)bn_max_blocks() * 2 * C;
  int nblk = 0;
  const void* mask = relu ? out : nullptr;
  const float slope = relu == 2 ? 0.01f : 0.f;  // activation code 2: leaky ReLU
  MI355_TRY(launch_bn_bwd_reduce(dtype, dout, mask, x, save_mean, save_invstd, dz_out, partial, &nblk, M, C, s, nullptr, slope));
  MI355_TRY(launch_bn_bwd_finalize(partial, nblk, M, C, gamma, save_invstd, dgamma, dbeta, beta_acc, coef, s));
  return launch_bn_bwd_apply(dtype, dout, mask, x, save_mean, save_invstd, coef, dx, M, C, s, nullptr, slope);
}

int mi355_maxpool_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  return launch_maxpool_fwd(dtype, x, y, idx, N, H, W, C, (hipStream_t)stream);
}
int mi355_maxpool_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                      void* stream) {
  return launch_maxpool_bwd(dtype, dy, idx, dx, N, H, W, C, (hipStream_t)stream);
}
int mi355_gap_fwd(int dtype, const void* x, float* pooled, int N, int HW, int C, void* stream) {
  return launch_gap_fwd(dtype, x, pooled, N, HW, C, (hipStream_t)stream);
}
int mi355_gap_bwd(int dtype, const float* dpooled, void* dx, int N, int HW, int C, void* stream) {
  return launch_gap_bwd(dtype, dpooled, dx, N, HW, C, (hipStream_t)stream);
}
int mi355_ce_loss(const float* logits, const float* target, float smoothing, float grad_scale, float* loss,
                  float* row_loss, float* dlogits, int N, int C, void* stream) {
  return launch_ce(logits, target, smoothing, grad_scale, loss, row_loss, dlogits, N, C, (hipStream_t)stream);
}
int mi355_sgd_step(float* p, const float* g, float* m, size_t n, float lr, float momentum, float weight_decay,
                   float grad_scale, void* stream) {
  return launch_sgd(p, g, m, n, lr, momentum, weight_decay, grad_scale, (hipStream_t)stream);
}
int mi355_sgd_step_ema(float* p, const float* g, float* m, float* ema, size_t n, float lr, float momentum, float weight_decay,
                       float grad_scale, float ema_decay, void* stream) {
  MI355_ARG(ema, "sgd_step_ema: null ema");
  return launch_sgd(p, g, m, n, lr, momentum, weight_decay, grad_scale, (hipStream_t)stream, ema, ema_decay);
}

}  // extern "C"
