// common.h — shared types/helpers for the gfx950 kernels of libmi355rn.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/mi355rn.h"

namespace mi355 {

typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

void set_error(const char* fmt, ...);
int device_cus();  // compute units of the current device (256 when none is visible): persistent grids = device_cus() x workgroups per CU
// value of a timing-probe environment variable (MI355_*_DBG), 0 when unset; the first non-zero read of each variable prints one
// line to stderr: with a probe active the kernels skip loads / stores and their RESULTS ARE WRONG (tools/probe8.py only)
int probe_env(const char* name);

#define MI355_HIP(expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      ::mi355::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));  \
      return MI355_E_HIP;                                                                       \
    }                                                                                           \
  } while (0)

#define MI355_LAUNCH_CHECK()                                                                    \
  do {                                                                                          \
    hipError_t e_ = hipGetLastError();                                                          \
    if (e_ != hipSuccess) {                                                                     \
      ::mi355::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return MI355_E_HIP;                                                                       \
    }                                                                                           \
  } while (0)

#define MI355_ARG(cond, ...)                      \
  do {                                            \
    if (!(cond)) {                                \
      ::mi355::set_error(__VA_ARGS__);            \
      return MI355_E_ARG;                         \
    }                                             \
  } while (0)

#define MI355_TRY(expr)            \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != 0) return rc_;      \
  } while (0)

static inline size_t dtype_size(int dt) { return dt == MI355_BF16 ? 2 : 4; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// Gather-GEMM geometry shared by conv forward and dgrad (conv_igemm.hip).
//   out[(n, i*OS+ph, j*OS+pw)][col] = sum_t sum_c in[(n, i*IS+dh_t, j*IS+dw_t)][c] * wt[col][wtap_t][c]
// (n,i,j) runs over a sub-grid N x Hsub x Wsub; rows whose source pixel is out of range read zeros.
// ---------------------------------------------------------------------------------------------
struct Tap {
  int8_t dh, dw;
  int16_t wtap;
};
struct TapClass {
  int ph, pw, ntaps;
  Tap taps[9];
};
struct IgemmArgs {
  const void* in;      // [N][Hin][Win] pixels of pix_stride elements
  const void* wt;      // [Ncols][wtaps][Ck], K-contiguous
  void* out;           // [N][Hout][Wout][Ncols]
  const void* addend;  // optional, laid out like out
  const uint8_t* addend_bits = nullptr;  // optional ReLU mask of the addend (1 byte per 16-byte vector): masked before the add
  int addend_sub2 = 0;  // 1: `addend` is a compact [N][Hout/2][Wout/2][Ncols] tensor standing for a full-resolution one whose odd rows /
                        // columns are zero (the data gradient of a stride-2 1x1 convolution, never written at full size); only the
                        // generated pointwise kernels take it (igemm_sub2_legal)
  int stat_rows_cap = 0;  // rows `stat_partial` can hold (0: the per-op API's 768); a launch never writes more partial rows than this
  float* stat_partial; // optional [stat_rows][2][Ncols]: per-workgroup-row sums of out and out^2 (BN statistics), or,
                       // when bn_y is set, of dz and dz*xhat (BN backward of the layer whose activation gradient `out`
                       // is): dz = out under the ReLU mask bn_bits, xhat = (bn_y - bn_mean) * bn_invstd
  void* sk_ws = nullptr;               // optional igemm_sk_ws_bytes() of zero-initialised scratch, owned by ONE stream:
                                       // lets the kernel cut the tiles of a partial last round along K (stream-K)
  const void* bn_y = nullptr;          // [N][Hout][Wout][Ncols], laid out like out
  const uint8_t* bn_bits = nullptr;    // 1 byte per 16-byte vector of out
  const float* bn_mean = nullptr;      // [Ncols]
  const float* bn_invstd = nullptr;    // [Ncols]
  float bn_slope = 0.f;                // 0: ReLU mask (dz = 0 where the bit is clear); 0.01: leaky ReLU (dz = out * 0.01 there) — generated bf16 kernels only (igemm_leaky_sums_legal)
  // BatchNorm + ReLU of the INPUT in the operand path (generated stride-1 3x3 and resident-weight 1x1 forward kernels only: igemm_bn_in_legal): `in` is the raw output y
  // of the previous convolution, the kernel reads relu(y * scale[c] + shift[c]) rounded to the tensor dtype — bn_apply's value — and leaves that
  // tensor and its ReLU bits in memory as a by-product (the weight gradient and the BN backward read them as before)
  const float* bn_in = nullptr;        // [2][Ck]: scale, shift
  void* bn_in_a = nullptr;             // laid out like in
  uint8_t* bn_in_bits = nullptr;       // 1 byte per 16-byte vector of it
  const float* q_scale_in = nullptr;   // fp8 operand launches (launch_igemm_fp8): the per-tensor quantisation scales of `in` and
  const float* q_scale_wt = nullptr;   // `wt` in device memory — out = acc / (*q_scale_in * *q_scale_wt); null: plain oscale
  int N, Hin, Win, pix_stride;
  int Hsub, Wsub, IS;
  int Hout, Wout, OS;
  int Ck, Ncols, wtaps;
  int pair_delta;      // stem only: elements to skip between k = 31 and k = 32 of a tap (the next image row); else 0
  TapClass cls[4];     // blockIdx.z selects the class
};

// Weight-gradient GEMM (conv_wgrad.hip):
//   partial[split][co][wtap_t][c] = sum_{m in split} dy[m][co] * x[(n, oh*IS+dh_t, ow*IS+dw_t)][c]
struct WgradArgs {
  const void* dy;   // [N][Ho][Wo][Cout]
  const void* x;    // [N][Hin][Win] pixels of pix_stride elements
  float* partial;   // [splits][Cout][wtaps][Ck]
  int N, Ho, Wo, Cout;
  int Hin, Win, pix_stride, IS;
  int Ck, wtaps, ntaps;
  int chunks_per_split;  // BKP-pixel chunks handled by one split
  int pair_delta;        // stem only: elements to skip between staged elements 31 and 32 of a tap (next image row); else 0
  Tap taps[9];
};

// ---- kernel launchers (all enqueue on `stream`, return mi355_status) -------------------------------
// stat_rows (optional): number of partial rows written to a.stat_partial, 0 if the statistics were not produced
// does a generated kernel with the leaky form of the BN-backward sums (IgemmArgs::bn_slope == 0.01) take this launch?  (no other kernel has that epilogue)
bool igemm_leaky_sums_legal(int dtype, const IgemmArgs& a, int nclass);
int launch_igemm(int dtype, const IgemmArgs& a, int nclass, hipStream_t stream, int* stat_rows = nullptr);
bool igemm_sub2_legal(int dtype, const IgemmArgs& a, int nclass);
bool dconv_fp8_legal(const IgemmArgs& a, int nclass);   // launch_igemm_fp8: the generated e4m3 3x3 kernel serves this launch
int launch_dconv_fp8(const IgemmArgs& a, int nclass, float oscale, hipStream_t stream, int* stat_rows);
bool igemm_bn_in_legal(int dtype, const IgemmArgs& a, int nclass);  // a launch with a.bn_in has a kernel (else: run bn_apply first)  // a launch with a.addend_sub2 has a kernel (else: materialise the addend)
static constexpr size_t IGEMM_SK_FLAG_BYTES = 4096;  // 512 flags + an error word, padded
static constexpr int IGEMM_SK_ERR_WORD = 512;        // index of the error word in the flag block: nonzero = a stream-K
                                                     // hand-off timed out in some launch that used this scratch
size_t igemm_sk_ws_bytes();
// fp8 (e4m3) operands in `a.in` / `a.wt` (1-byte elements), bf16 output: the 8-wave kernel with 128-channel k-tiles (fp8.hip picks
// the tile).  igemm_fp8_legal: geometry the kernel can run (Ck % 128, Ncols % 128, ...).
bool igemm_fp8_legal(const IgemmArgs& a, int nclass);
int launch_igemm_fp8(const IgemmArgs& a, int nclass, float oscale, hipStream_t stream, int* stat_rows = nullptr);
// splits chosen by plan_wgrad_splits(); partial must hold splits*Cout*wtaps*Ck floats
int plan_wgrad_splits(int dtype, int M, int Cout, int ntaps, int Ck);
// plan_wgrad: the split count for THIS launch — the generated 3x3 kernels (asm/wg_gen.py, dconv.cpp) have their own, every other
// launch plan_wgrad_splits' — and what launch_wgrad must be given for the launch to take the kernel the plan was made for
// Per-launch tile knobs of the conv launchers (test hooks and A/B switches: MI355_IGEMM8, MI355_IGEMM_BIG, MI355_STEM_DIRECT, MI355_STEM_TH,
// MI355_STEM_DBG): read from the environment ONCE and on mi355_reload_knobs() — no getenv on a launch path.
struct Knobs {
  bool has_igemm8 = false;
  char igemm8[32] = {0};     // MI355_IGEMM8: "0" never, "<BM>x<BN>[k][f]" forces a tile; unset: the measured rule
  int igemm_big = -1;        // MI355_IGEMM_BIG: -1 unset, 0 never, 1 256 x 256 wherever legal, 3 256 x 128
  bool has_igemm_big = false;
  int stem_direct = 1;       // MI355_STEM_DIRECT=0: the row-pair implicit GEMM
  int stem_th = 0;           // MI355_STEM_TH: a smaller stem tile (0: the plan)
  bool stem_dbg = false;     // MI355_STEM_DBG: print the stem launch plan
  bool sk_mute = false;      // MI355_SK_DEBUG=mute: stream-K contributors never publish (test hook: the owner's bounded wait must time out and report)
  // kernel-selection switches of the generated gfx950 kernels (A/B): "0" keeps the launches they serve on the implicit-GEMM kernels
  int dconv = 1;             // MI355_DCONV: every generated kernel
  int wg3 = 1;               // MI355_WG3: the generated weight-gradient kernels (asm/wg_gen.py, wg1_gen.py)
  int pw = 1;                // MI355_PW: the persistent pointwise kernel (asm/pw_gen.py)
  int pk = 1;                // MI355_PK: the long-reduction pointwise kernels (asm/pk_gen.py)
  int po = 1;                // MI355_PO: the output-heavy pointwise kernels with resident weights (asm/po_gen.py); 1: the measured per-shape rule, 2: wherever a variant is legal
  int po64 = 1;              // MI355_PO64: the 64-column forms of po (layer 1's 1x1 launches into 64 channels): 0 leaves them on the implicit-GEMM kernel,
                             // 1: the measured per-shape rule, 2: wherever legal
  int dconv_fp8 = 1;         // MI355_DCONV_FP8: the e4m3 step's stride-1 3x3 launches on the generated kernels (dconv_*_q: K = 128 MFMA); 0: the 8-wave HIP kernel
  int pool_keys = 1;         // MI355_POOL_KEYS: the stem's BN + ReLU + max pool on packed (value, tap) keys (misc.hip bn_relu_maxpool3_kernel; 0: the compare / select form)
  int po_bn = 1;             // MI355_PO_BN: bn2 + ReLU in conv3's operand path (po_*_s1_a0_bn: HBM-bound launches, the transform hides under the memory time)
  int dconv_bn = 0;          // MI355_DCONV_BN=1: bn1 + ReLU in conv2's operand path (dconv_*_s1_bn: the executor's training forward skips that bn_apply launch).
                             // Off by default: measured break-even (profiles/r06_ab_bn_in_operand_path.txt: the transform's VALU work is not hidden in a
                             // one-wave-per-SIMD MFMA-bound kernel: +8 .. +30 us per launch against 8 .. 37 us of bn_apply)
  int dconv_s2 = 1;          // MI355_DCONV_S2: the generated kernels of the stride-2 3x3 convolutions (dconv_*_d2: the data gradient by output-parity classes)
  char error[160] = {0};     // a switch with a value outside its domain: every conv launch fails with MI355_E_ARG and this text
};
const Knobs& knobs();
// an executor switch, read at context creation: unset -> def; one digit "0" .. maxv -> that value; anything else -> *bad = the switch's name
// (the creation then fails with MI355_E_ARG: a stale or mistyped switch in a job script is an error, not a silent default)
inline int env_switch(const char* name, int def, int maxv, const char** bad) {
  const char* e = getenv(name);
  if (!e) return def;
  if (e[0] >= '0' && e[0] <= '0' + maxv && e[1] == 0) return e[0] - '0';
  if (bad && !*bad) *bad = name;
  return def;
}
int plan_wgrad(int dtype, const WgradArgs& a);
int wg3_plan(int dtype, const WgradArgs& a);  // 0: the launch is not served by a generated kernel
// name of the kernel the last conv / weight-gradient launch of this thread went to (a generated kernel's symbol, or the implicit-GEMM
// tile family): tests assert the selection rules with it (mi355_last_conv_kernel)
void note_kernel(const char* fmt, ...);
int launch_wg3(const WgradArgs& a, int splits, hipStream_t stream);
int launch_wgrad(int dtype, const WgradArgs& a, int splits, hipStream_t stream);
// dst[i] = beta*dst[i] + sum_s partial[s][i], i < n  (n multiple of 4); deterministic order
// sa / sb (optional device scalars, fp8 wgrad): the sum is multiplied by 1 / (*sa * *sb) before beta * dst is added
int launch_splitk_reduce(const float* partial, int splits, size_t stride, float* dst, size_t n, float beta,
                         hipStream_t stream, const float* sa = nullptr, const float* sb = nullptr);
// stem unpack: dw[64][7][7][3] = beta*dw + sum_s partial[s][co][kh>>1][(kh&1)*32 + kw*4+c]
int launch_stem_unpack(float* partial, int splits, float* dw, float beta, hipStream_t stream);  // reduces `partial` in place

// weight preparation (weights.hip): fp32 KRSC master -> what the kernels consume
//   wT (dtype) [Cout][taps][Cin]   (skipped when dtype==F32: the master is used directly; pass NULL)
//   wTt(dtype) [Cin][taps][Cout]   transposed for dgrad (NULL to skip)
int launch_weight_prep(int dtype, const float* w, void* w_cast, void* w_tr, int Cout, int taps, int Cin,
                       hipStream_t stream);
// the same for every conv layer of a network in one launch; `table` (device memory) is sorted by tile_begin
constexpr int PREP_TILE = 64;  // channels per side of a weight-prep tile (every conv of the network has multiples of 64)
struct PrepDesc {
  size_t w_off;    // offset of the fp32 master in `params` (elements)
  void* w_cast;    // or null (cast copy not needed: fp32 ctx)
  void* w_tr;      // or null (no dgrad: inference forward)
  void* w_q;       // fp8 ctx, layers whose forward runs on e4m3 operands: [Cout][taps][Cin] bytes = e4m3(bf16(w) * *q_scale), or null
  void* w_trq;     // ... and the transposed [Cin][taps][Cout] bytes for dgrad, or null
  const float* q_scale;  // device scalar used this step
  unsigned* q_amax;      // device scalar: max |bf16(w)| as float bits, atomicMax'ed (the next step's scale)
  int Cout, taps, Cin;
  int tile_begin;  // first block of this layer; a layer has (Cout/PREP_TILE)*(Cin/PREP_TILE)*taps blocks
};
// BResNet-50 executor (variant.hip): [weight standardisation] -> zero pad -> cast, and the transposed copy, of EVERY convolution in two launches
// (one workgroup per padded output row, then 32 x 32 transpose tiles) instead of three small launches per layer — 164 launches whose dispatch
// latency, not their bytes, delayed the forward's second convolution.  Same arithmetic and summation order as mi355_weight_std_fwd /
// launch_weight_pad_cast / launch_transpose_any (the per-op graph's path): bit-identical weights.
struct BPrepDesc {
  const float* w;    // fp32 master [Cout][taps][Cin]
  float* w_hat;      // standardised fp32 copy (read by the weight gradient's backward), or null: standardisation off
  float* mean;       // [Cout]
  float* invstd;
  void* wp;          // [Coutp][taps][Cinp] in the compute dtype
  void* wtr;         // [Cinp][taps][Coutp], or null (inference forward)
  int Cout, taps, Cin, Coutp, Cinp;
  int row_begin;     // first workgroup of this layer in the row launch (Coutp workgroups)
  int tile_begin;    // ... in the transpose launch ((Cinp / 32) * (Coutp / 32) * taps workgroups)
};
// drop-connect / dropout scale arrays of one forward pass sampled by ONE launch (the values of mi355_keep_scale for the same seed and counter)
struct KeepBatch {
  float* keep[20];
  unsigned long long n[20];
  unsigned long long counter[20];
  float p[20];
  int count = 0;
};
int launch_keep_scale_batch(const KeepBatch& kb, unsigned long long seed, hipStream_t s);
int launch_bres_weight_prep(int dtype, const BPrepDesc* table, int nconv, int total_rows, int total_tiles, float eps, bool transposed, hipStream_t stream);
int launch_weight_prep_batch(int dtype, const PrepDesc* table, int nlayers, int total_tiles, const float* params,
                             hipStream_t stream);
// same-dtype transpose [Cout][taps][Cin] -> [Cin][taps][Cout] (per-op API, weights already in `dtype`)
int launch_transpose_any(int dtype, const void* w, void* wt, int Cout, int taps, int Cin, hipStream_t stream);
// stem: w[64][7][7][3] fp32 -> packed [64][4][64] dtype: row pair kh>>1, element (kh&1)*32 + kw*4+c, zero padded
int launch_stem_pack(int dtype, const float* w, void* packed, hipStream_t stream);

// conv geometry builders (conv_api.cpp)
void build_fwd_args(IgemmArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int build_dgrad_args(IgemmArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
void build_wgrad_args(WgradArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
void build_stem_fwd_args(IgemmArgs& a, int N, int H, int W);
void build_stem_wgrad_args(WgradArgs& a, int N, int H, int W);

// stem geometry: padded NHWC4 input
static constexpr int STEM_PAD = 3, STEM_PS = 4, STEM_CK = 64, STEM_RPAD = 16;
static inline int stem_hp(int H) { return H + 2 * STEM_PAD; }
static inline int stem_wp(int W) { return W + STEM_RPAD; }  // 2*(W/2-1)+16 = W+14 <= Wp, Wp even

// BN (bn.hip)
int launch_bn_stats(int dtype, const void* x, float* partial, float* pivot, int* nblk_out, int M, int C, hipStream_t s);
int launch_bn_finalize(const float* partial, const float* pivot, int nblk, int M, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                       float* scale, float* shift, float eps, float momentum, hipStream_t s);
int launch_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float* scale,
                          float* shift, int C, float eps, hipStream_t s);
// out = act(x*scale+shift (+ residual) (+ x2*scale2+shift2))
// Quantised twin of an activation / gradient tensor (fp8 training step): q = e4m3(bf16(v) * *scale) beside the bf16 tensor the
// kernel writes anyway, and max |bf16(v)| atomicMax'ed into *amax (float bits; seeds the NEXT step's scale).  q null: off.
struct QuantOut {
  uint8_t* q = nullptr;
  const float* scale = nullptr;
  unsigned* amax = nullptr;
  bool only = false;  // every consumer of the tensor reads the twin: the bf16 tensor itself is not written
};
int launch_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                    const void* x2, const float* scale2, const float* shift2, void* out, int M, int C, int relu,
                    hipStream_t s, uint8_t* relu_bits = nullptr, QuantOut qo = QuantOut());
// relu_bits: the ReLU mask as one byte per 16-byte vector of the tensor (bit e = element e of the vector was > 0);
// written by launch_bn_apply, read by the two backward kernels in place of the post-activation tensor
// optional gradient source of the BN-backward passes: g stands for g * keep[n] * gate[n][c] + dpool[n][c], n = pixel / hw — the backward of an
// ECA module (channel gate + drop-connect scale) applied on the fly (the BResNet-50 executor: the module's input gradient is never stored)
struct EcaGrad {
  const float* keep = nullptr;   // [N] or null
  const float* gate = nullptr;   // [N][C]
  const float* dpool = nullptr;  // [N][C]
  int hw = 0;
};
int launch_bn_bwd_reduce(int dtype, const void* g, const void* mask_src, const void* x, const float* mean,
                         const float* invstd, void* dz_out, float* partial, int* nblk_out, int M, int C,
                         hipStream_t s, const uint8_t* relu_bits = nullptr, float slope = 0.f, const EcaGrad* eg = nullptr);
int launch_bn_bwd_finalize(const float* partial, int nblk, int M, int C, const float* gamma, const float* invstd,
                           float* dgamma, float* dbeta, float beta_acc, float* coef /*[3][C]*/, hipStream_t s);
int launch_bn_bwd_apply(int dtype, const void* g, const void* mask_src, const void* x, const float* mean,
                        const float* invstd, const float* coef, void* dx, int M, int C, hipStream_t s,
                        const uint8_t* relu_bits = nullptr, float slope = 0.f, QuantOut qo = QuantOut(), const EcaGrad* eg = nullptr);
// conv + BN + ReLU + maxpool3x3/2 stage (the stem): both BN-backward passes gather the pool's backward from the pooled gradient
// dp [N][H/2][W/2][C] and the argmax codes; the full-resolution pool gradient is never stored (H, W: the full resolution)
int launch_stem_bwd_reduce(int dtype, const void* dp, const uint8_t* idx, const uint8_t* bits, const void* y, const float* mean,
                           const float* invstd, float* partial, int* nblk_out, int N, int H, int W, int C, hipStream_t s);
int launch_stem_bwd_apply(int dtype, const void* dp, const uint8_t* idx, const uint8_t* bits, const void* y, const float* mean,
                          const float* invstd, const float* coef, void* dx, int N, int H, int W, int C, hipStream_t s);
int bn_max_blocks();
// variant.hip: glue kernels of the static BResNet-50 executor
int launch_nchw_pad64(int dtype, const float* x_nchw, void* h_nhwc64, int N, int HW, hipStream_t s);
int launch_weight_pad_cast(int dtype, const float* w, void* wp, int Cout, int taps, int Cin, int Coutp, int Cinp, hipStream_t s);
int launch_nchw_im2col3s2(int dtype, const float* x, void* h, int N, int H, int W, hipStream_t s);  // 3x3 / stride-2 patches of an NCHW fp32 batch, 27 of 64 channels
int launch_weight_unpad(const float* dwp, float* dw, float beta, int Cout, int taps, int Cin, int Cinp, hipStream_t s);
int launch_axpby(const float* src, float* dst, float beta, size_t n, hipStream_t s);
// xs / xh (optional): x stands for x * xs[c] + xh[c] (a BatchNorm with identity activation applied on the fly); ss / sh2: the same for the shortcut
int launch_eca_residual_fwd(int dtype, const void* x, const float* w, int k, const float* keep, const void* shortcut, void* out, float* pooled, float* gate,
                            int N, int HW, int C, int act, hipStream_t s, const float* xs = nullptr, const float* xh = nullptr, const float* ss = nullptr,
                            const float* sh2 = nullptr, uint8_t* out_bits = nullptr, float* raw_ws = nullptr /* [N][C] scratch: xs != null -> the pooled means' affine + the gate in one launch */);
int launch_eca_residual_bwd(int dtype, const void* dout, const void* out, const void* x, const float* keep, const float* w, int k, const float* pooled,
                            const float* gate, void* dshortcut, void* dx, float* dw, float beta, float* ws, int N, int HW, int C, int act, hipStream_t s,
                            const float* xs = nullptr, const float* xh = nullptr, float* bn_row = nullptr, const float* bn_mean = nullptr,
                            const float* bn_invstd = nullptr, const uint8_t* out_bits = nullptr, const void* ds_y = nullptr, float* ds_row = nullptr,
                            const float* ds_mean = nullptr, const float* ds_invstd = nullptr);
// fp8 step: scale[i] = amax[i] > 0 ? 448 / (headroom * amax[i]) : scale[i];  amax[i] = 0   (delayed per-tensor scaling)
int launch_fp8_scale_update(float* scale, unsigned* amax, int n, float headroom, hipStream_t s);

// pooling / head / loss / optimizer
int launch_maxpool_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, hipStream_t s);
// p = maxpool3x3/2(relu(y*scale+shift)) + argmax + the ReLU bit mask of the (never stored) full-resolution activation
int launch_bn_relu_maxpool(int dtype, const void* y, const float* scale, const float* shift, void* p, uint8_t* idx,
                           uint8_t* bits, int N, int H, int W, int C, hipStream_t s);
int launch_maxpool_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                       hipStream_t s);
int launch_gap_fwd(int dtype, const void* x, float* pooled, int N, int HW, int C, hipStream_t s);
int launch_fc(const float* x, const float* w, float* out, int M, int N, int K, hipStream_t s);  // fc.hip: out = x * w^T, fp32
int launch_gap_bwd(int dtype, const float* dpooled, void* dx, int N, int HW, int C, hipStream_t s);
int launch_ce(const float* logits, const float* target, float smoothing, float grad_scale, float* loss,
              float* row_loss, float* dlogits, int N, int C, hipStream_t s);
int launch_sgd(float* p, const float* g, float* m, size_t n, float lr, float mom, float wd, float gscale,
               hipStream_t s, float* ema = nullptr, float ema_decay = 0.f);
int launch_stem_ingest(int dtype, const float* x, void* xpad, int N, int H, int W, hipStream_t s);
// logits[n][o] = tmp[n*ld + o] + bias[o]
int launch_bias_slice(const float* tmp, int ld, const float* bias, float* out, int N, int O, hipStream_t s);
// dst[n*ld + o] = o < O ? src[n*O + o] : 0 ; dbias[o] = beta*dbias + sum_n src[n][o]
int launch_pad_dlogits(const float* src, float* dst, int ld, float* dbias, float beta, int N, int O, hipStream_t s);

}  // namespace mi355
