// resnet_exec.cpp — static ResNet-50 v1.5 training/inference executor behind the C-ABI (include/mi355rn.h).
//
// Replaces what the reference gets from `hydra.utils.call(cfg.model)` -> pytorch_tools.models.resnet50
// (train.py:64, configs/hydra_exp/1.r50_baseline.yaml:22-23) plus the autograd graph torch builds under it
// (model(data) / loss.backward(): sota_imagenet/callbacks.py:316-317).  The whole forward and backward
// schedule is fixed at ctx creation: one workspace arena, NHWC activations, no allocation and no host
// synchronisation inside a step.
//
// Schedule (torchvision layout): stem 7x7/2 -> BN -> ReLU -> maxpool 3x3/2 -> 16 bottlenecks
// (1x1 -> 3x3(stride) -> 1x1, BN after each, ReLU after the first two, residual add + ReLU at the end,
// 1x1(stride)+BN downsample on the first block of each stage) -> global average pool -> FC.
#include <cstdlib>
#include <string>
#include <vector>

#include "comm.h"
#include "common.h"

namespace mi355 {

void build_stem_fwd_args(IgemmArgs& a, int N, int H, int W);
void build_stem_wgrad_args(WgradArgs& a, int N, int H, int W);

namespace {

constexpr float BN_EPS = 1e-5f;
constexpr float FP8_HEADROOM = 2.f;  // scale = 448 / (FP8_HEADROOM * amax of the previous step): one binade of the e4m3 range
constexpr size_t PARAM_ALIGN = 64;  // floats (256 B)
constexpr int MAX_EVENTS = 8192;

struct TensorInfo {
  std::string name;
  int kind;  // 0 param, 1 buffer
  size_t offset;
  int ndim;
  int shape[4];
};

struct ConvBN {
  std::string conv_name, bn_name;
  int Cin, Cout, K, stride, pad;
  int Hin, Win, Hout, Wout;
  size_t w_off = 0, gamma_off = 0, beta_off = 0;  // flat params
  size_t rm_off = 0, rv_off = 0;                  // flat buffers
  void* y = nullptr;                              // raw conv output
  void* w_cast = nullptr;                         // bf16 copy (bf16 ctx)
  void* w_tr = nullptr;                           // transposed copy for dgrad
  float* stat = nullptr;                          // [4][Cout]: save_mean, save_invstd, scale, shift
  int splits = 1;   // split count of the weight-gradient launch (plan_wgrad)
  int splits8 = 1;  // ... of its e4m3 form (fp8 step: the implicit-GEMM kernel with its own plan)
  int stat_rows = 0;  // partial rows the last forward conv left in bn_partial (0: none)
  int bwd_rows = 0;   // partial rows of THIS layer's BN-backward sums left in bn_partial by the dgrad that produced its
                      // activation gradient (0: none, bn_backward runs the standalone reduce kernel)
  bool is_stem = false;
  // fp8 training step (ctx dtype MI355_FP8): forward / dgrad of this layer on e4m3 operands where the geometry allows it
  bool fp8_fwd = false, fp8_dgrad = false, fp8_wgrad = false;
  void* w_q = nullptr;          // e4m3 [Cout][taps][Cin]
  void* w_trq = nullptr;        // e4m3 [Cin][taps][Cout]
  const void* in_q = nullptr;   // e4m3 twin of the layer's input activation (written by the bn_apply that produced it)
  int qid_w = -1, qid_in = -1, qid_dy = -1;  // slots of the per-tensor scale / amax tables
  // the kernels this layer's last forward / data-gradient / weight-gradient launch went to (mi355_resnet50_kernel_table)
  char k_fwd[48] = "", k_dgrad[48] = "", k_wgrad[48] = "";
};

struct Block {
  ConvBN c1, c2, c3, ds;
  bool has_ds = false;
  const void* in = nullptr;  // block input activation
  void* a1 = nullptr;
  void* a2 = nullptr;
  void* out = nullptr;
  uint8_t *a1_bits = nullptr, *a2_bits = nullptr, *out_bits = nullptr;  // ReLU masks, 1 byte per 16-byte vector
  uint8_t *a1_q = nullptr, *a2_q = nullptr, *out_q = nullptr;           // e4m3 twins (fp8 step; null where no fp8 conv reads them)
  int qid_a1 = -1, qid_a2 = -1, qid_out = -1;
  int Hin, Win, Hout, Wout, Cin, Cout;
  size_t grad_begin = 0, grad_end = 0;
};

struct Arena {
  size_t size = 0;
  std::vector<std::pair<void**, size_t>> slots;
  void add(void** p, size_t bytes) {
    slots.push_back({p, size});
    size += align_up(bytes, 256);
  }
};

}  // namespace
}  // namespace mi355

using namespace mi355;

constexpr int MAX_GSETS = 4;

struct mi355_ctx {
  int device = 0, dtype = 0, N = 0, H = 0, W = 0, num_classes = 0, fc_pad = 0;
  size_t es = 4;
  ConvBN stem;
  std::vector<Block> blocks;
  std::vector<TensorInfo> tensors;
  size_t param_elems = 0, buffer_elems = 0;
  size_t fc_w_off = 0, fc_b_off = 0;
  size_t fc_grad_begin = 0, fc_grad_end = 0, stem_grad_begin = 0, stem_grad_end = 0;
  float *params = nullptr, *grads = nullptr, *buffers = nullptr;

  // workspace
  char* arena = nullptr;
  size_t arena_bytes = 0;
  void *xpad = nullptr, *stem_pack = nullptr, *p0 = nullptr;
  uint8_t* pool_idx = nullptr;
  uint8_t* a0_bits = nullptr;
  float *pooled = nullptr, *fc_tmp = nullptr, *dlogits_pad = nullptr, *dpooled = nullptr, *fc_wtr = nullptr;
  float *bn_partial = nullptr, *bn_coef = nullptr, *wg_partial = nullptr;
  float *bn_partial2 = nullptr, *bn_coef2 = nullptr;  // the same for BN work issued to the side stream
  void* sk_ws[2] = {nullptr, nullptr};                // stream-K scratch of the conv kernel, main / side stream
  bool stream_k = true;
  // gradient collective inside the boundary (comm.cpp): buckets of consecutive backward segments, each reduced by one
  // mean all-reduce on the communicator's stream as soon as its last segment has been enqueued
  mi355_comm* comm = nullptr;
  struct Bucket {
    size_t begin, end;
    int last_seg;
  };
  std::vector<Bucket> buckets;
  // fp8 training step: bf16 tensors everywhere + e4m3 twins of the conv operands; delayed per-tensor scaling (a tensor's amax of
  // step k sets its scale of step k + 1); the first training step of a ctx runs bf16 operands and only records the amaxes
  bool fp8 = false;
  int q_n = 0;
  float* q_scale = nullptr;
  unsigned* q_amax = nullptr;
  void* gq[MAX_GSETS][4] = {};  // e4m3 twins of gset
  bool fp8_fwd_cal = false, fp8_bwd_cal = false;  // a training forward / a whole backward has recorded its amaxes
  bool fp8_fwd_on = false, fp8_bwd_on = false;    // this step's convs read the e4m3 twins
  bool fp8_use_fwd = true, fp8_use_bwd = true;    // MI355_FP8_FWD=0 / MI355_FP8_BWD=0 (read at create): keep that direction on bf16 operands (A/B)
  bool fp8_use_wgrad = true;                      // MI355_FP8_WGRAD=0: weight gradients stay on the bf16 tensors
  bool fp8_keep_bf16 = false;                     // MI355_FP8_KEEP_BF16=1: write the bf16 tensors even where every consumer reads the twin
                                                  // (tests compare the twins with them)
  bool fp8_lean = false;                          // this step: bf16 tensors whose consumers all read twins are not written
  bool grad_sync = true;    // false: backward skips the bucket all-reduces (DDP.no_sync(): non-final accumulation micro-steps)
  bool comm_dirty = false;  // an all-reduce of this backward call is in flight on the communicator's stream
  unsigned* sk_err_host = nullptr;  // pinned copy of the two scratch blocks' error words, refreshed by an async copy at the
                                    // end of every forward / backward call and looked at (no wait) at the start of the next
  PrepDesc* prep_table[2] = {nullptr, nullptr};  // [0]: cast only (inference), [1]: cast + transposed (training)
  int prep_layers = 0, prep_tiles = 0;
  // backward gradients: gG[2] carry the gradient wrt a block output down the network; gset[p] holds the per-layer
  // gradients (dy3, dy_ds, dy2, dy1) of the blocks of parity p — two sets so that the weight-gradient stream may
  // still be reading block k's gradients while block k+1 is written
  void* gG[2] = {nullptr, nullptr};
  void* gsub = nullptr;    // the downsample branch's data gradient at HALF resolution (first blocks of layers 2-4): conv1's data gradient
                           // adds it at the even pixels (IgemmArgs::addend_sub2), the zero-filled full-size tensor is never written
  bool ds_compact = true;  // MI355_DS_COMPACT=0: the stride-2 data gradient written at full size (A/B; the fp32 path always does)
  void* gset[MAX_GSETS][4] = {};
  size_t max_c = 64;  // widest BatchNorm of the network (sizes bn_partial)
  int nsets = 2;  // gradient buffer sets in rotation (MI355_GSETS, read at ctx creation): the main stream waits for the weight gradients of
                  // the block that used a set nsets blocks ago
  // weight-gradient side stream (wgrad + split-K reduce run beside the BN-backward / dgrad chain of the main stream)
  bool overlap = false;
  bool fuse_bn_bwd = false;  // BN-backward sums in the dgrad epilogues (MI355_FUSE_BN_BWD=0/1 overrides the default)
  bool fuse_bn_in = true;    // bn1 + ReLU in conv2's operand path where a generated kernel has that form (library switch MI355_DCONV_BN)
  bool stem_fused_bwd = true;  // stem BN backward gathers the pool gradient on the fly (MI355_STEM_FUSED=0: pool-backward kernel + plain BN backward)
  hipStream_t wstream = nullptr;
  std::vector<hipEvent_t> fork_ev;
  size_t fork_next = 0;
  hipEvent_t w_done[MAX_GSETS] = {};
  hipEvent_t ds_done = nullptr;  // the downsample branch issued to the side stream has finished
  bool w_pending[MAX_GSETS] = {};
  bool w_dirty = false;
  int bwd_parity = 0;
  bool fwd_training_done = false;
  int next_seg = 0;
  void* cur_dout = nullptr;  // gradient wrt the current block output during backward
  double fwd_flops = 0, train_flops = 0;

  // profiling
  unsigned prof_mask = 0;
  std::vector<hipEvent_t> ev;
  struct Rec {
    int cls;
    int tag;  // bit 0: a 3x3 convolution (profile_read kind 8 = the 3x3 launches of classes 0 and 1); bit 1: e4m3 operands (kind 9)
    double flops, bytes;
  };
  std::vector<Rec> recs;
};

namespace {

size_t act_bytes(const mi355_ctx* c, int H, int W, int C) { return (size_t)c->N * H * W * C * c->es; }

void add_param(mi355_ctx* c, std::vector<TensorInfo>& rev, const std::string& name, size_t* off, int ndim, int s0,
               int s1 = 0, int s2 = 0, int s3 = 0, size_t extra_pad = 0) {
  size_t n = (size_t)s0 * (ndim > 1 ? s1 : 1) * (ndim > 2 ? s2 : 1) * (ndim > 3 ? s3 : 1);
  *off = c->param_elems;
  c->param_elems += align_up(n + extra_pad, PARAM_ALIGN);
  TensorInfo t{name, 0, *off, ndim, {s0, s1, s2, s3}};
  rev.push_back(t);
}
void add_buffer(mi355_ctx* c, std::vector<TensorInfo>& rev, const std::string& name, size_t* off, int n) {
  *off = c->buffer_elems;
  c->buffer_elems += align_up((size_t)n, PARAM_ALIGN);
  TensorInfo t{name, 1, *off, 1, {n, 0, 0, 0}};
  rev.push_back(t);
}

void init_conv(ConvBN& l, const std::string& conv_name, const std::string& bn_name, int Cin, int Cout, int K,
               int stride, int Hin, int Win) {
  l.conv_name = conv_name;
  l.bn_name = bn_name;
  l.Cin = Cin; l.Cout = Cout; l.K = K; l.stride = stride; l.pad = K / 2;
  l.Hin = Hin; l.Win = Win;
  l.Hout = (Hin + 2 * l.pad - K) / stride + 1;
  l.Wout = (Win + 2 * l.pad - K) / stride + 1;
}

// params of one conv+bn in reverse-execution registration order: bn.bias, bn.weight, conv.weight
void register_convbn(mi355_ctx* c, std::vector<TensorInfo>& rev, ConvBN& l) {
  add_param(c, rev, l.bn_name + ".bias", &l.beta_off, 1, l.Cout);
  add_param(c, rev, l.bn_name + ".weight", &l.gamma_off, 1, l.Cout);
  add_param(c, rev, l.conv_name + ".weight", &l.w_off, 4, l.Cout, l.Cin, l.K, l.K);
  add_buffer(c, rev, l.bn_name + ".running_var", &l.rv_off, l.Cout);
  add_buffer(c, rev, l.bn_name + ".running_mean", &l.rm_off, l.Cout);
}

struct Prof {
  mi355_ctx* c;
  hipStream_t stream;
  int idx = -1;
  Prof(mi355_ctx* ctx, int cls, double flops, double bytes, hipStream_t s, int tag = 0) : c(ctx), stream(s) {
    if (!(c->prof_mask & (1u << cls))) return;
    if ((c->recs.size() + 1) * 2 > c->ev.size()) return;
    idx = (int)c->recs.size();
    c->recs.push_back({cls, tag, flops, bytes});
    (void)hipEventRecord(c->ev[2 * idx], stream);
  }
  ~Prof() {
    if (idx >= 0) (void)hipEventRecord(c->ev[2 * idx + 1], stream);
  }
};

// profile classes
// profile classes = kernel symbols (so HIP-event averages can be checked against rocprofv3 --stats)
enum { PC_IGEMM128 = 0, PC_IGEMM64 = 1, PC_WGRAD128 = 2, PC_WGRAD64 = 3, PC_BN_REDUCE = 4, PC_BN_APPLY = 5, PC_OTHER = 6,
       PC_BN_BWD_APPLY = 7 };
inline int igemm_class(int ncols) { return ncols % 128 == 0 ? PC_IGEMM128 : PC_IGEMM64; }
inline int wgrad_class(int cout) { return cout % 128 == 0 ? PC_WGRAD128 : PC_WGRAD64; }

// BN scratch of the stream the work is issued to (the side stream has its own, the two run concurrently)
inline float* bn_partial_of(mi355_ctx* c, hipStream_t s) { return (c->wstream && s == c->wstream) ? c->bn_partial2 : c->bn_partial; }
inline void* sk_ws_of(mi355_ctx* c, hipStream_t s) {
  return c->stream_k ? c->sk_ws[(c->wstream && s == c->wstream) ? 1 : 0] : nullptr;
}
inline float* bn_coef_of(mi355_ctx* c, hipStream_t s) { return (c->wstream && s == c->wstream) ? c->bn_coef2 : c->bn_coef; }
double conv_flops(const mi355_ctx* c, const ConvBN& l) {
  return 2.0 * c->N * l.Hout * l.Wout * (double)l.Cout * l.Cin * l.K * l.K;
}

// conv forward; in training the epilogue also leaves the BN statistics partials of its output in c->bn_partial
// (l.stat_rows > 0), unless the shape does not allow it (then bn_prepare runs the standalone statistics kernel)
// bn_in (optional): the conv+BN layer whose RAW output `in` is — its BatchNorm + ReLU is applied in this conv's operand path and the
// activation + its ReLU bits land in a_out / a_bits as a by-product (the caller has checked conv_bn_in_legal)
int conv_forward(mi355_ctx* c, ConvBN& l, const void* in, int training, float momentum, hipStream_t s, const ConvBN* bn_in = nullptr,
                 void* a_out = nullptr, uint8_t* a_bits = nullptr) {
  IgemmArgs a;
  build_fwd_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
  a.in = in;
  if (bn_in) {
    a.bn_in = bn_in->stat + 2 * bn_in->Cout;  // scale[C], shift[C]: contiguous in the layer's statistics block
    a.bn_in_a = a_out;
    a.bn_in_bits = a_bits;
  }
  a.stat_partial = training ? bn_partial_of(c, s) : nullptr;
  a.stat_rows_cap = (int)((size_t)bn_max_blocks() * c->max_c / l.Cout);  // the scratch holds bn_max_blocks() rows of the widest layer
  a.sk_ws = sk_ws_of(c, s);
  a.wt = c->dtype == MI355_F32 ? (const void*)(c->params + l.w_off) : (const void*)l.w_cast;
  a.out = l.y;
  const double fl = conv_flops(c, l);
  const bool q = training && c->fp8_fwd_on && l.fp8_fwd;
  const double by = (double)c->N * l.Hin * l.Win * l.Cin * (q ? 1 : c->es) * (bn_in ? 2.0 + 1.0 / 16 : 1.0) + (double)c->N * l.Hout * l.Wout * l.Cout * c->es;
  Prof p(c, igemm_class(l.Cout), fl, by, s, (l.K == 3 ? 1 : 0) | (q ? 2 : 0));
  if (q) {
    a.in = l.in_q; a.wt = l.w_q;
    a.q_scale_in = c->q_scale + l.qid_in; a.q_scale_wt = c->q_scale + l.qid_w;
    MI355_TRY(launch_igemm_fp8(a, 1, 1.f, s, &l.stat_rows));
  } else {
    MI355_TRY(launch_igemm(c->dtype, a, 1, s, &l.stat_rows));
  }
  snprintf(l.k_fwd, sizeof(l.k_fwd), "%s", mi355_last_conv_kernel());
  return 0;
}

// conv `l` can take the BatchNorm + ReLU of its producer `prev` in its operand path (training forward, bf16 operands, a generated kernel with that form:
// conv2 <- bn1 on the direct 3x3 kernels, conv3 <- bn2 on the resident-weight pointwise kernels)
bool conv_bn_in_legal(mi355_ctx* c, const ConvBN& prev, const ConvBN& l, int training) {
  if (!training || c->dtype != MI355_BF16 || c->fp8 || !c->fuse_bn_in) return false;
  IgemmArgs a;
  build_fwd_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
  a.in = prev.y; a.wt = l.w_cast; a.out = l.y;
  a.stat_partial = c->bn_partial;
  a.stat_rows_cap = (int)((size_t)bn_max_blocks() * c->max_c / l.Cout);
  a.bn_in = prev.stat; a.bn_in_a = prev.y; a.bn_in_bits = (uint8_t*)prev.y;  // (placeholders: only their presence matters for the check)
  return igemm_bn_in_legal(c->dtype, a, 1);
}

// BN statistics + finalize for one layer (training) or eval coefficients
int bn_prepare(mi355_ctx* c, ConvBN& l, int training, float momentum, hipStream_t s) {
  const int M = c->N * l.Hout * l.Wout, C = l.Cout;
  float* save_mean = l.stat;
  float* save_invstd = l.stat + C;
  float* scale = l.stat + 2 * C;
  float* shift = l.stat + 3 * C;
  const float* gamma = c->params + l.gamma_off;
  const float* beta = c->params + l.beta_off;
  if (training) {
    int nblk = l.stat_rows;
    const float* pivot = nullptr;  // conv-epilogue partials are plain sums
    if (nblk == 0) {
      Prof p(c, PC_BN_REDUCE, 0, (double)M * C * c->es, s);
      MI355_TRY(launch_bn_stats(c->dtype, l.y, bn_partial_of(c, s), bn_coef_of(c, s), &nblk, M, C, s));
      pivot = bn_coef_of(c, s);
    }
    return launch_bn_finalize(bn_partial_of(c, s), pivot, nblk, M, C, gamma, beta, c->buffers + l.rm_off, c->buffers + l.rv_off,
                              save_mean, save_invstd, scale, shift, BN_EPS, momentum, s);
  }
  return launch_bn_eval_coeffs(gamma, beta, c->buffers + l.rm_off, c->buffers + l.rv_off, scale, shift, C, BN_EPS, s);
}

int bn_apply(mi355_ctx* c, ConvBN& l, const void* residual, ConvBN* l2, void* out, int relu, hipStream_t s,
             uint8_t* bits, uint8_t* q = nullptr, int qid = -1, bool q_only = false) {
  const int M = c->N * l.Hout * l.Wout, C = l.Cout;
  const int nin = 1 + (residual ? 1 : 0) + (l2 ? 1 : 0);
  QuantOut qo;
  if (q && bits && qid >= 0) {  // training forward of the fp8 step: the e4m3 twin the next conv(s) read + its amax
    qo.q = q; qo.scale = c->q_scale + qid; qo.amax = c->q_amax + qid;
    qo.only = q_only && !residual && !l2;
  }
  Prof p(c, PC_BN_APPLY, 0, (double)M * C * (c->es * (nin + (qo.only ? 0 : 1)) + (qo.q ? 1 : 0)), s);
  return launch_bn_apply(c->dtype, l.y, l.stat + 2 * C, l.stat + 3 * C, residual, l2 ? l2->y : nullptr,
                         l2 ? l2->stat + 2 * C : nullptr, l2 ? l2->stat + 3 * C : nullptr, out, M, C, relu, s, bits, qo);
}

// e4m3 twin of one of the per-layer gradient buffers (gset[p][i] -> gq[p][i]); null for any other pointer
void* grad_twin(const mi355_ctx* c, const void* g) {
  for (int p = 0; p < c->nsets; ++p)
    for (int i = 0; i < 4; ++i)
      if (c->gset[p][i] == g) return c->gq[p][i];
  return nullptr;
}

// BN backward of layer l: g (gradient wrt the activation), bits (ReLU mask of the activation, 1 byte per 16-byte
// vector, or null), optional in-place masked write-back; dx written to `dx` (may alias g).
int bn_backward(mi355_ctx* c, ConvBN& l, const void* g, const uint8_t* bits, void* dz_out, void* dx, float beta_acc,
                hipStream_t s) {
  const int M = c->N * l.Hout * l.Wout, C = l.Cout;
  int nblk = l.bwd_rows;
  l.bwd_rows = 0;
  const double mask_bytes = bits ? (double)M * C * c->es / 16 : 0.0;
  if (nblk == 0 || dz_out) {
    Prof p(c, PC_BN_REDUCE, 0, (double)M * C * c->es * (2 + (dz_out ? 1 : 0)) + mask_bytes, s);
    MI355_TRY(launch_bn_bwd_reduce(c->dtype, g, nullptr, l.y, l.stat, l.stat + C, dz_out, bn_partial_of(c, s), &nblk, M, C, s,
                                   bits, 0.f));
  }
  MI355_TRY(launch_bn_bwd_finalize(bn_partial_of(c, s), nblk, M, C, c->params + l.gamma_off, l.stat + C,
                                     c->grads + l.gamma_off, c->grads + l.beta_off, beta_acc, bn_coef_of(c, s), s));
  // after an in-place masked write-back the mask is already applied
  const uint8_t* bits2 = dz_out ? nullptr : bits;
  const void* g2 = dz_out ? dz_out : g;
  QuantOut qo;
  if (c->fp8 && l.qid_dy >= 0) {  // dx = the gradient wrt this layer's conv output: the operand of its fp8 dgrad / wgrad
    qo.q = (uint8_t*)grad_twin(c, dx); qo.scale = c->q_scale + l.qid_dy; qo.amax = c->q_amax + l.qid_dy;
    // both consumers (dgrad, wgrad) read the twin: the bf16 gradient is not written
    qo.only = c->fp8_lean && c->fp8_bwd_on && l.fp8_dgrad && l.fp8_wgrad && qo.q;
  }
  Prof p(c, PC_BN_BWD_APPLY, 0, (double)M * C * (c->es * (qo.only ? 2 : 3) + (qo.q ? 1 : 0)) + (bits2 ? mask_bytes : 0.0), s);
  return launch_bn_bwd_apply(c->dtype, g2, nullptr, l.y, l.stat, l.stat + C, bn_coef_of(c, s), dx, M, C, s, bits2, 0.f, qo);
}

int conv_wgrad(mi355_ctx* c, ConvBN& l, const void* dy, const void* x, float beta_acc, hipStream_t s) {
  WgradArgs a;
  build_wgrad_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
  a.dy = dy; a.x = x; a.partial = c->wg_partial;
  const size_t n = (size_t)l.Cout * l.K * l.K * l.Cin;
  void* dyq = c->fp8_bwd_on && l.fp8_wgrad ? grad_twin(c, dy) : nullptr;
  const double by = ((double)c->N * l.Hin * l.Win * l.Cin + (double)c->N * l.Hout * l.Wout * l.Cout) * (dyq ? 1 : c->es);
  if (dyq) {  // e4m3 twins of both operands; the sums are rescaled by 1 / (scale_dy * scale_x) in the split reduce
    a.dy = dyq; a.x = l.in_q;
    {
      Prof p(c, wgrad_class(l.Cout), conv_flops(c, l), by, s, 2);
      MI355_TRY(launch_wgrad(MI355_FP8, a, l.splits8, s));
    }
    snprintf(l.k_wgrad, sizeof(l.k_wgrad), "%s", mi355_last_conv_kernel());
    return launch_splitk_reduce(c->wg_partial, l.splits8, n, c->grads + l.w_off, n, beta_acc, s, c->q_scale + l.qid_dy, c->q_scale + l.qid_in);
  }
  {
    Prof p(c, wgrad_class(l.Cout), conv_flops(c, l), by, s);
    MI355_TRY(launch_wgrad(c->dtype, a, l.splits, s));
  }
  snprintf(l.k_wgrad, sizeof(l.k_wgrad), "%s", mi355_last_conv_kernel());
  return launch_splitk_reduce(c->wg_partial, l.splits, n, c->grads + l.w_off, n, beta_acc, s);
}

// dx = dgrad(dy) (+ addend under its mask).  bn (optional): the conv+BN layer whose post-ReLU activation dx is the
// gradient of — the epilogue then also leaves that layer's BN-backward sums in bn_partial (bn->bwd_rows), which saves
// the standalone reduce pass over dx and bn->y.
// sub2: `addend` is the half-resolution tensor conv_dgrad_compact() left (IgemmArgs::addend_sub2)
int conv_dgrad(mi355_ctx* c, ConvBN& l, const void* dy, void* dx, const void* addend, hipStream_t s,
               const uint8_t* addend_bits = nullptr, ConvBN* bn = nullptr, const uint8_t* bn_bits = nullptr, float beta_acc = 0.f, bool sub2 = false) {
  IgemmArgs a;
  const int nclass = build_dgrad_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
  if (nclass < 0) return nclass;
  a.in = dy; a.wt = l.w_tr; a.out = dx; a.addend = addend; a.addend_bits = addend_bits; a.addend_sub2 = sub2 ? 1 : 0;
  a.sk_ws = sk_ws_of(c, s);
  if (bn && c->fuse_bn_bwd) {
    a.stat_partial = bn_partial_of(c, s);
    a.stat_rows_cap = (int)((size_t)bn_max_blocks() * c->max_c / bn->Cout);
    a.bn_y = bn->y; a.bn_bits = bn_bits; a.bn_mean = bn->stat; a.bn_invstd = bn->stat + bn->Cout;
  }
  const double dx_elems = (double)c->N * l.Hin * l.Win * l.Cin;
  void* dyq = c->fp8_bwd_on && l.fp8_dgrad ? grad_twin(c, dy) : nullptr;
  // dy read, dx written, + the addend and (fused BN-backward sums) that layer's y read in the epilogue, masks at 1/16
  const double by = dx_elems * (1 + (addend ? (sub2 ? 0.25 : 1) : 0) + (a.bn_y ? 1 : 0)) * c->es + (double)c->N * l.Hout * l.Wout * l.Cout * (dyq ? 1 : c->es) +
                    dx_elems * c->es / 16 * ((addend_bits ? 1 : 0) + (a.bn_y ? 1 : 0));
  Prof p(c, igemm_class(l.Cin), conv_flops(c, l), by, s, (l.K == 3 ? 1 : 0) | (dyq ? 2 : 0));
  int* rows = bn && c->fuse_bn_bwd ? &bn->bwd_rows : nullptr;
  if (dyq) {
    a.in = dyq; a.wt = l.w_trq;
    a.q_scale_in = c->q_scale + l.qid_dy; a.q_scale_wt = c->q_scale + l.qid_w;
    MI355_TRY(launch_igemm_fp8(a, nclass, 1.f, s, rows));
  } else {
    MI355_TRY(launch_igemm(c->dtype, a, nclass, s, rows));
  }
  snprintf(l.k_dgrad, sizeof(l.k_dgrad), "%s", mi355_last_conv_kernel());
  return 0;
}

// the stride-2 1x1 downsample convolution's data gradient at the OUTPUT resolution: dxs[n][ho][wo] = dy[n][ho][wo] * W^T, i.e. the only
// pixels of the full-size gradient that are not zero, as a dense 1x1 launch (the generated pointwise kernels instead of a four-class
// implicit GEMM that writes 3/4 zeros)
int conv_dgrad_compact(mi355_ctx* c, ConvBN& l, const void* dy, void* dxs, hipStream_t s) {
  IgemmArgs a;
  const int nclass = build_dgrad_args(a, c->N, l.Hout, l.Wout, l.Cin, l.Cout, 1, 1, 1, 0);
  if (nclass < 0) return nclass;
  a.in = dy; a.wt = l.w_tr; a.out = dxs;
  const double px = (double)c->N * l.Hout * l.Wout;
  Prof p(c, igemm_class(l.Cin), 2.0 * px * l.Cin * l.Cout, px * (l.Cin + l.Cout) * c->es, s, 0);
  MI355_TRY(launch_igemm(c->dtype, a, nclass, s, nullptr));
  snprintf(l.k_dgrad, sizeof(l.k_dgrad), "%s", mi355_last_conv_kernel());
  return 0;
}

// conv1's data gradient of a first block can take the downsample gradient at half resolution (a generated kernel serves that launch)
bool sub2_ok(mi355_ctx* c, Block& b, Block* prev) {
  if (!c->ds_compact || c->dtype != MI355_BF16 || !b.has_ds || b.ds.stride != 2 || b.ds.K != 1) return false;
  if (c->fp8 && b.ds.fp8_dgrad) return false;  // (the e4m3 step keeps the downsample data gradient on bf16 operands for this: plan_fp8)
  IgemmArgs a;
  ConvBN& l = b.c1;
  const int nclass = build_dgrad_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
  if (nclass < 0) return false;
  a.in = a.wt = a.addend = (const void*)1;  // (geometry only: the legality test looks at which pointers are set, not where they point)
  a.out = (void*)1;
  a.addend_sub2 = 1;
  if (prev && c->fuse_bn_bwd) {
    a.stat_partial = (float*)1; a.bn_y = (const void*)1; a.bn_bits = (const uint8_t*)1; a.bn_mean = a.bn_invstd = (const float*)1;
    a.stat_rows_cap = (int)((size_t)bn_max_blocks() * c->max_c / prev->c3.Cout);
  }
  return igemm_sub2_legal(c->dtype, a, nclass);
}

int plan_arena(mi355_ctx* c, Arena& ar) {
  const int N = c->N;
  size_t max_act = 0, max_wg = 0, max_c = 64;  // (max_c is kept in the ctx: conv launches derive their partial-row capacity from it)
  auto conv_ws = [&](ConvBN& l) {
    ar.add(&l.y, act_bytes(c, l.Hout, l.Wout, l.Cout));
    ar.add((void**)&l.stat, (size_t)4 * l.Cout * 4);
    max_act = std::max(max_act, act_bytes(c, l.Hout, l.Wout, l.Cout));
    max_act = std::max(max_act, act_bytes(c, l.Hin, l.Win, l.Cin));
    max_c = std::max<size_t>(max_c, l.Cout);
    if (!l.is_stem) {
      const size_t wn = (size_t)l.Cout * l.K * l.K * l.Cin;
      if (c->dtype != MI355_F32) ar.add(&l.w_cast, wn * c->es);
      ar.add(&l.w_tr, wn * c->es);
      if (l.qid_w >= 0) {
        ar.add(&l.w_q, wn);
        ar.add(&l.w_trq, wn);
      }
      WgradArgs wa;
      build_wgrad_args(wa, N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
      // (the e4m3 twins of the fp8 step run on the implicit-GEMM kernel with its own plan; the bf16 launches of the same ctx —
      // calibration step, layers kept in bf16 — take the plan, and so the kernels and the bits, of the bf16 step)
      l.splits = plan_wgrad(c->dtype, wa);
      l.splits8 = c->fp8 ? plan_wgrad_splits(c->dtype, N * l.Hout * l.Wout, l.Cout, l.K * l.K, l.Cin) : l.splits;
      max_wg = std::max(max_wg, (size_t)std::max(l.splits, l.splits8) * wn * 4);
    } else {
      l.splits = plan_wgrad_splits(c->dtype, N * l.Hout * l.Wout, 64, 4, STEM_CK);
      max_wg = std::max(max_wg, (size_t)l.splits * 64 * 4 * 64 * 4);
    }
  };
  ar.add(&c->xpad, mi355_stem_xpad_bytes(c->dtype, N, c->H, c->W));
  ar.add(&c->stem_pack, (size_t)64 * 4 * 64 * c->es);
  conv_ws(c->stem);
  ar.add((void**)&c->a0_bits, act_bytes(c, c->stem.Hout, c->stem.Wout, 64) / 16);
  ar.add(&c->p0, act_bytes(c, c->stem.Hout / 2, c->stem.Wout / 2, 64));
  ar.add((void**)&c->pool_idx, (size_t)N * (c->stem.Hout / 2) * (c->stem.Wout / 2) * 64);
  for (auto& b : c->blocks) {
    conv_ws(b.c1);
    ar.add(&b.a1, act_bytes(c, b.c1.Hout, b.c1.Wout, b.c1.Cout));
    ar.add((void**)&b.a1_bits, act_bytes(c, b.c1.Hout, b.c1.Wout, b.c1.Cout) / 16);
    conv_ws(b.c2);
    ar.add(&b.a2, act_bytes(c, b.c2.Hout, b.c2.Wout, b.c2.Cout));
    ar.add((void**)&b.a2_bits, act_bytes(c, b.c2.Hout, b.c2.Wout, b.c2.Cout) / 16);
    conv_ws(b.c3);
    if (b.has_ds) conv_ws(b.ds);
    ar.add(&b.out, act_bytes(c, b.Hout, b.Wout, b.Cout));
    ar.add((void**)&b.out_bits, act_bytes(c, b.Hout, b.Wout, b.Cout) / 16);
    if (b.qid_a1 >= 0) ar.add((void**)&b.a1_q, act_bytes(c, b.c1.Hout, b.c1.Wout, b.c1.Cout) / c->es);
    if (b.qid_a2 >= 0) ar.add((void**)&b.a2_q, act_bytes(c, b.c2.Hout, b.c2.Wout, b.c2.Cout) / c->es);
    if (b.qid_out >= 0) ar.add((void**)&b.out_q, act_bytes(c, b.Hout, b.Wout, b.Cout) / c->es);
  }
  const int fcp = c->fc_pad;
  ar.add((void**)&c->pooled, (size_t)N * 2048 * 4);
  ar.add((void**)&c->fc_tmp, (size_t)N * fcp * 4);
  ar.add((void**)&c->dlogits_pad, (size_t)N * fcp * 4);
  ar.add((void**)&c->dpooled, (size_t)N * 2048 * 4);
  ar.add((void**)&c->fc_wtr, (size_t)fcp * 2048 * 4);
  {
    const int fsplits = plan_wgrad_splits(MI355_F32, N, fcp, 1, 2048);
    max_wg = std::max(max_wg, (size_t)fsplits * fcp * 2048 * 4);
  }
  for (int i = 0; i < 2; ++i) ar.add((void**)&c->prep_table[i], (size_t)64 * sizeof(PrepDesc));
  c->max_c = max_c;
  ar.add((void**)&c->bn_partial, (size_t)bn_max_blocks() * 2 * max_c * 4);
  ar.add((void**)&c->bn_coef, (size_t)3 * max_c * 4);
  for (int i = 0; i < 2; ++i) ar.add(&c->sk_ws[i], igemm_sk_ws_bytes());
  ar.add((void**)&c->bn_partial2, (size_t)bn_max_blocks() * 2 * max_c * 4);
  ar.add((void**)&c->bn_coef2, (size_t)3 * max_c * 4);
  ar.add((void**)&c->wg_partial, max_wg);
  for (int i = 0; i < 2; ++i) ar.add(&c->gG[i], max_act);
  ar.add(&c->gsub, max_act / 4);
  for (int p = 0; p < c->nsets; ++p)
    for (int i = 0; i < 4; ++i) ar.add(&c->gset[p][i], max_act);
  if (c->fp8) {
    for (int p = 0; p < c->nsets; ++p)
      for (int i = 0; i < 4; ++i) ar.add(&c->gq[p][i], max_act / c->es);
    ar.add((void**)&c->q_scale, (size_t)c->q_n * 4);
    ar.add((void**)&c->q_amax, (size_t)c->q_n * 4);
  }
  return 0;
}

// fp8 step: which layers run on e4m3 operands (geometry the fp8 form of the 8-wave kernel accepts: channel counts in multiples of
// 128 on both sides — layers 2-4 of the network; stem, layer 1 and FC stay bf16), the twins they read, and their scale slots.
void plan_fp8(mi355_ctx* c, const char** bad) {
  // MI355_FP8_PLAN=rule: the per-layer exclusions of mid round 5 below.  Default since the K = 128 / K = 64 matrix instructions (conv_igemm8.hip, conv_wgrad.hip):
  // every legal launch on e4m3 operands — 33.98 against 34.54 ms per step at batch 512 (two alternations on one box): single launches the rule excluded are
  // still a few us slower than the generated bf16 kernel (3x3 weight gradients 98-111 against 88-99 us, layer 3's conv3 forward 84-88 / 65-75), but with
  // every consumer of a tensor on its e4m3 twin the bf16 copy is not written at all, and the step follows its bytes
  const char* pa = getenv("MI355_FP8_PLAN");   // "all" (default) | "rule"
  if (pa && strcmp(pa, "all") != 0 && strcmp(pa, "rule") != 0 && bad && !*bad) *bad = "MI355_FP8_PLAN";
  const bool plan_all = !(pa && pa[0] == 'r');
  const bool ds_compact_env = env_switch("MI355_DS_COMPACT", 1, 1, bad) != 0;
  int n = 0;
  // Measured per layer at batch 256 / 224 px (profiles/r03a_conv_per_layer_fp8_vs_bf16.txt): the e4m3 form wins 4-10 us per launch
  // wherever the reduction spans >= 2 of its 128-channel k-tiles, and loses 15-90 us on the output-heavy launches — the 128 -> 512
  // 1x1 forward of layer 2 (ONE k-tile under a 512-column epilogue) and every conv1 dgrad (K = 128 ... 512 under the shortcut
  // addend + BN-backward epilogue: the single workgroup per CU of the 8-wave kernel runs that epilogue with the matrix pipe idle,
  // the same reason the bf16 rule keeps them on the 4-wave kernel).  Those stay bf16 and get no twins.
  auto legal = [&](ConvBN& l, bool is_c1) {
    IgemmArgs a;
    build_fwd_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
    l.fp8_fwd = igemm_fp8_legal(a, 1) && l.Cin * l.K * l.K >= 256;
    const int nclass = build_dgrad_args(a, c->N, l.Hin, l.Win, l.Cin, l.Cout, l.K, l.K, l.stride, l.pad);
    l.fp8_dgrad = nclass > 0 && igemm_fp8_legal(a, nclass) && !is_c1;
    // Round 5: re-derived against the GENERATED bf16 kernels (profiles/r05_config5_conv_per_layer_{bf16,fp8}_224.txt, batch 512): the e4m3
    // implicit-GEMM kernel loses to the direct 3x3 kernel of layer 2 (forward 114 against 139 us, data gradient 141 / 159) and to the
    // resident-weight pointwise kernel on layer 3's conv3 forward (71 / 92); it still wins on layer 4's 3x3 (86 / 114), on every long
    // 1x1 reduction (conv1 forward of layers 2-4, conv3's data gradient) and on the stride-2 / downsample launches of the first blocks.
    if (!plan_all) {
      if (l.K == 3 && l.stride == 1 && l.Cin == 128) l.fp8_fwd = l.fp8_dgrad = false;
      if (l.K == 1 && l.stride == 1 && l.Cin == 256 && l.Cout == 1024) l.fp8_fwd = false;
    }
    // the downsample conv's data gradient stays on bf16 operands: dense at half resolution + conv1's data gradient reading it there
    // (conv_dgrad_compact / sub2_ok) beats the four-class e4m3 launch at full size by 0.49 ms per step at batch 512 (148 / 283, 444 / 609,
    // 113 / 182, 222 / 311, 160 / 194 us: tools/fp8_plan_table.sh)
    if (ds_compact_env && l.K == 1 && l.stride == 2) l.fp8_dgrad = false;
    if (l.fp8_fwd || l.fp8_dgrad) l.qid_w = n++;
    if (l.fp8_dgrad) l.qid_dy = n++;
  };
  c->fp8_use_wgrad = true;
  c->fp8_keep_bf16 = env_switch("MI355_FP8_KEEP_BF16", 0, 1, bad) != 0;
  for (size_t i = 0; i < c->blocks.size(); ++i) {
    Block& b = c->blocks[i];
    legal(b.c1, true); legal(b.c2, false); legal(b.c3, false);
    if (b.has_ds) legal(b.ds, false);
    if (i > 0 && (b.c1.fp8_fwd || (b.has_ds && b.ds.fp8_fwd))) {
      Block& prev = c->blocks[i - 1];
      prev.qid_out = n++;
      b.c1.qid_in = b.c1.fp8_fwd ? prev.qid_out : -1;
      b.ds.qid_in = b.has_ds && b.ds.fp8_fwd ? prev.qid_out : -1;
    } else {
      b.c1.fp8_fwd = false;  // (the first block reads the max-pool output, which has no twin)
      b.ds.fp8_fwd = false;
    }
    if (b.c2.fp8_fwd) b.c2.qid_in = b.qid_a1 = n++;
    if (b.c3.fp8_fwd) b.c3.qid_in = b.qid_a2 = n++;
    // weight gradients on e4m3 operands (transposed byte reads, conv_wgrad.hip ES = 1) wherever the layer's input has a twin anyway
    // (its forward reads it) and both channel counts are multiples of 128; the gradient twin is then written for conv1 too
    auto wg_plan = [&](ConvBN& l) {
      l.fp8_wgrad = c->fp8_use_wgrad && l.fp8_fwd && l.qid_in >= 0 && l.Cin % 128 == 0 && l.Cout % 128 == 0;
      // (round 5: the generated bf16 weight-gradient kernels beat the e4m3 implicit-GEMM form on every stride-1 3x3 — 83-94 against
      // 115-126 us — and on layer 4's 1x1s, 50 / 60; the e4m3 form keeps layer 2's conv1, 70 / 110, and the first blocks)
      if (!plan_all) {
        if (l.K == 3 && l.stride == 1) l.fp8_wgrad = false;
        if (l.K == 1 && l.stride == 1 && l.Cin * l.Cout == 2048 * 512) l.fp8_wgrad = false;
      }
      if (l.fp8_wgrad && l.qid_dy < 0) l.qid_dy = n++;
    };
    wg_plan(b.c1); wg_plan(b.c2); wg_plan(b.c3);
    if (b.has_ds) wg_plan(b.ds);
  }
  c->q_n = (int)align_up((size_t)n, 64);
}

int weight_prep_all(mi355_ctx* c, bool need_tr, hipStream_t s) {
  Prof p(c, PC_OTHER, 0, 0, s);
  MI355_TRY(launch_stem_pack(c->dtype, c->params + c->stem.w_off, c->stem_pack, s));
  if (need_tr || c->dtype != MI355_F32)
    MI355_TRY(launch_weight_prep_batch(c->dtype, c->prep_table[need_tr ? 1 : 0], c->prep_layers, c->prep_tiles, c->params, s));
  if (need_tr)
    MI355_TRY(launch_weight_prep(MI355_F32, c->params + c->fc_w_off, nullptr, c->fc_wtr, c->fc_pad, 1, 2048, s));
  return 0;
}

// ---- weight-gradient side stream ----------------------------------------------------------------------------------
// fork(): the stream the next wgrad goes to, ordered after everything issued to `s` so far.
int fork(mi355_ctx* c, hipStream_t s, hipStream_t* w) {
  if (!c->overlap) {
    *w = s;
    return 0;
  }
  hipEvent_t e = c->fork_ev[c->fork_next++ % c->fork_ev.size()];
  MI355_HIP(hipEventRecord(e, s));
  MI355_HIP(hipStreamWaitEvent(c->wstream, e, 0));
  c->w_dirty = true;
  *w = c->wstream;
  return 0;
}
// before the main stream overwrites gradient set p: the wgrads of the block that used it last must have read it
int acquire_set(mi355_ctx* c, int p, hipStream_t s) {
  if (c->overlap && c->w_pending[p]) {
    MI355_HIP(hipStreamWaitEvent(s, c->w_done[p], 0));
    c->w_pending[p] = false;
  }
  return 0;
}
int release_set(mi355_ctx* c, int p) {
  if (c->overlap) {
    MI355_HIP(hipEventRecord(c->w_done[p], c->wstream));
    c->w_pending[p] = true;
  }
  return 0;
}
// join(): everything on the side stream becomes visible to `s` (end of a backward call: the caller reads the gradients)
int join(mi355_ctx* c, hipStream_t s) {
  if (!c->overlap || !c->w_dirty) return 0;
  hipEvent_t e = c->fork_ev[c->fork_next++ % c->fork_ev.size()];
  MI355_HIP(hipEventRecord(e, c->wstream));
  MI355_HIP(hipStreamWaitEvent(s, e, 0));
  c->w_dirty = false;
  for (int p = 0; p < MAX_GSETS; ++p) c->w_pending[p] = false;
  return 0;
}

// stream-K (fp32 conv): a hand-off that timed out leaves a wrong tile behind; the kernel raises an error word, which is
// copied to pinned host memory behind the kernels of a call and checked at the start of the following calls
int sk_check(mi355_ctx* c) {
  if (c->sk_err_host && (c->sk_err_host[0] | c->sk_err_host[1])) {
    set_error("a stream-K hand-off of an earlier convolution launch timed out (partial tile lost): the results of that "
              "step are invalid; recreate the context (MI355_STREAM_K=0 disables the tile cutting)");
    return MI355_E_STATE;
  }
  return 0;
}
int sk_snapshot(mi355_ctx* c, hipStream_t s) {
  if (!c->sk_err_host || !c->stream_k) return 0;
  for (int i = 0; i < 2; ++i)
    if (c->sk_ws[i])
      MI355_HIP(hipMemcpyAsync(c->sk_err_host + i, reinterpret_cast<unsigned*>(c->sk_ws[i]) + IGEMM_SK_ERR_WORD, sizeof(unsigned),
                               hipMemcpyDeviceToHost, s));
  return 0;
}

void seg_range(const mi355_ctx* c, int seg, size_t* b, size_t* e) {
  const int nb = (int)c->blocks.size();
  if (seg == 0) {
    *b = c->fc_grad_begin; *e = c->fc_grad_end;
  } else if (seg == nb + 1) {
    *b = c->stem_grad_begin; *e = c->stem_grad_end;
  } else {
    const Block& blk = c->blocks[nb - seg];
    *b = blk.grad_begin; *e = blk.grad_end;
  }
}
// consecutive segments (backward completion order = ascending offsets) form buckets of >= cap_elems gradient elements.
// The LAST bucket has nothing left to hide behind (its all-reduce starts when backward ends), so it is cut once more: its trailing
// segments up to cap_elems / 8 (stem + layer 1 + the end of layer 2: ~4 MB at the default cap) become a bucket of their own and the
// part before them is reduced while those last, activation-heavy blocks are still computing.
std::vector<mi355_ctx::Bucket> plan_buckets(const mi355_ctx* c, size_t cap_elems) {
  std::vector<mi355_ctx::Bucket> out;
  const int nseg = (int)c->blocks.size() + 2;
  bool open = false;
  size_t start = 0;
  int first_seg = 0;
  std::vector<int> firsts;
  for (int i = 0; i < nseg; ++i) {
    size_t b, e;
    seg_range(c, i, &b, &e);
    if (!open) {
      start = b;
      first_seg = i;
      open = true;
    }
    if (e - start >= cap_elems || i == nseg - 1) {
      out.push_back({start, e, i});
      firsts.push_back(first_seg);
      open = false;
    }
  }
  const size_t tail_cap = cap_elems / 8;
  if (!out.empty() && out.back().end - out.back().begin > tail_cap) {
    const int f = firsts.back(), l = out.back().last_seg;
    int cut = l + 1;  // first segment of the tail bucket (l + 1: no tail — the last segment alone exceeds the tail cap)
    size_t tail = 0;
    for (int i = l; i > f; --i) {
      size_t b, e;
      seg_range(c, i, &b, &e);
      if (tail + (e - b) > tail_cap) break;
      tail += e - b;
      cut = i;
    }
    if (cut > f && cut <= l) {
      size_t b, e;
      seg_range(c, cut, &b, &e);
      const mi355_ctx::Bucket last = out.back();
      out.back() = {last.begin, b, cut - 1};
      out.push_back({b, last.end, l});
    }
  }
  return out;
}

int backward_fc(mi355_ctx* c, const float* dlogits, float beta_acc, hipStream_t s) {
  const int N = c->N, O = c->num_classes, P = c->fc_pad;
  MI355_TRY(launch_pad_dlogits(dlogits, c->dlogits_pad, P, c->grads + c->fc_b_off, beta_acc, N, O, s));
  // wgrad: dW[o][k] = sum_n dlogits[n][o] * pooled[n][k]
  WgradArgs w;
  build_wgrad_args(w, N, 1, 1, 2048, P, 1, 1, 1, 0);
  w.dy = c->dlogits_pad; w.x = c->pooled; w.partial = c->wg_partial;
  const int splits = plan_wgrad_splits(MI355_F32, N, P, 1, 2048);
  hipStream_t ws;
  MI355_TRY(fork(c, s, &ws));
  MI355_TRY(launch_wgrad(MI355_F32, w, splits, ws));
  MI355_TRY(launch_splitk_reduce(c->wg_partial, splits, (size_t)P * 2048, c->grads + c->fc_w_off, (size_t)O * 2048,
                                 beta_acc, ws));
  // dgrad: dpooled[n][k] = sum_o dlogits[n][o] * W[o][k]
  MI355_TRY(launch_fc(c->dlogits_pad, c->fc_wtr, c->dpooled, N, 2048, P, s));
  const Block& last = c->blocks.back();
  c->cur_dout = c->gG[0];
  c->bwd_parity = 0;
  return launch_gap_bwd(c->dtype, c->dpooled, c->cur_dout, N, last.Hout * last.Wout, last.Cout, s);
}

int backward_block(mi355_ctx* c, Block& b, Block* prev, float beta_acc, hipStream_t s) {  // prev: the block below
  void* G = c->cur_dout;  // gradient wrt the block output, BEFORE its ReLU mask (b.out_bits): the mask is applied on the
                          // fly by the three consumers (bn3 / downsample-bn backward, the shortcut add of conv1's dgrad)
  void* Gn = G == c->gG[0] ? c->gG[1] : c->gG[0];
  const int par = c->bwd_parity;
  c->bwd_parity = (par + 1) % c->nsets;
  void** S = c->gset[par];
  void *B1 = S[0], *B2 = S[1], *B3 = S[2], *B4 = S[3];
  hipStream_t ws;
  const bool sub2 = sub2_ok(c, b, prev);
  MI355_TRY(acquire_set(c, par, s));
  MI355_TRY(bn_backward(c, b.c3, G, b.out_bits, nullptr, B1, beta_acc, s));  // B1 = dy3
  MI355_TRY(fork(c, s, &ws));
  MI355_TRY(conv_wgrad(c, b.c3, B1, b.a2, beta_acc, ws));
  if (b.has_ds) {
    // the whole downsample branch runs beside the conv3 -> conv1 chain (same fork: it only needs G)
    MI355_TRY(bn_backward(c, b.ds, G, b.out_bits, nullptr, B2, beta_acc, ws));  // B2 = dyd
    if (sub2) MI355_TRY(conv_dgrad_compact(c, b.ds, B2, c->gsub, ws));           // gsub = shortcut gradient at half resolution
    else MI355_TRY(conv_dgrad(c, b.ds, B2, Gn, nullptr, ws));                   // Gn = shortcut gradient
    if (c->overlap) MI355_HIP(hipEventRecord(c->ds_done, ws));
    MI355_TRY(conv_wgrad(c, b.ds, B2, b.in, beta_acc, ws));
  }
  MI355_TRY(conv_dgrad(c, b.c3, B1, B3, nullptr, s, nullptr, &b.c2, b.a2_bits, beta_acc));  // B3 = da2 (+ bn2's sums)
  MI355_TRY(bn_backward(c, b.c2, B3, b.a2_bits, nullptr, B3, beta_acc, s));  // B3 = dy2
  MI355_TRY(fork(c, s, &ws));
  MI355_TRY(conv_wgrad(c, b.c2, B3, b.a1, beta_acc, ws));
  MI355_TRY(conv_dgrad(c, b.c2, B3, B4, nullptr, s, nullptr, &b.c1, b.a1_bits, beta_acc));  // B4 = da1 (+ bn1's sums)
  MI355_TRY(bn_backward(c, b.c1, B4, b.a1_bits, nullptr, B4, beta_acc, s));  // B4 = dy1
  MI355_TRY(fork(c, s, &ws));
  MI355_TRY(conv_wgrad(c, b.c1, B4, b.in, beta_acc, ws));
  if (b.has_ds) {
    if (c->overlap) MI355_HIP(hipStreamWaitEvent(s, c->ds_done, 0));
    // Gn = dx_in = conv1 dgrad + shortcut gradient (+ the sums of the previous block's bn3)
    MI355_TRY(conv_dgrad(c, b.c1, B4, Gn, sub2 ? c->gsub : Gn, s, nullptr, prev ? &prev->c3 : nullptr, prev ? prev->out_bits : nullptr, beta_acc, sub2));
    c->cur_dout = Gn;
  } else {
    // G = dx_in = conv1 dgrad + masked G (+ the sums of the previous block's bn3)
    MI355_TRY(conv_dgrad(c, b.c1, B4, G, G, s, b.out_bits, prev ? &prev->c3 : nullptr, prev ? prev->out_bits : nullptr, beta_acc));
    c->cur_dout = G;
  }
  return release_set(c, par);
}

int backward_stem(mi355_ctx* c, float beta_acc, hipStream_t s) {
  void* G = c->cur_dout;  // gradient wrt maxpool output
  const int par = c->bwd_parity;
  c->bwd_parity = (par + 1) % c->nsets;
  void* B1 = c->gset[par][0];
  ConvBN& l = c->stem;
  MI355_TRY(acquire_set(c, par, s));
  if (c->stem_fused_bwd) {
    // B1 = dy of the stem conv straight from the pooled gradient: the 4x larger pool gradient is never written / re-read
    const int M = c->N * l.Hout * l.Wout, C = l.Cout;
    const double full = (double)M * C * c->es, pooled = full / 4 + (double)M * C / 4, mask = full / 16;
    int nblk = 0;
    {
      Prof p(c, PC_BN_REDUCE, 0, full + pooled + mask, s);
      MI355_TRY(launch_stem_bwd_reduce(c->dtype, G, c->pool_idx, c->a0_bits, l.y, l.stat, l.stat + C, bn_partial_of(c, s), &nblk, c->N, l.Hout,
                                       l.Wout, C, s));
    }
    MI355_TRY(launch_bn_bwd_finalize(bn_partial_of(c, s), nblk, M, C, c->params + l.gamma_off, l.stat + C, c->grads + l.gamma_off,
                                     c->grads + l.beta_off, beta_acc, bn_coef_of(c, s), s));
    Prof p(c, PC_BN_BWD_APPLY, 0, 2 * full + pooled + mask, s);
    MI355_TRY(launch_stem_bwd_apply(c->dtype, G, c->pool_idx, c->a0_bits, l.y, l.stat, l.stat + C, bn_coef_of(c, s), B1, c->N, l.Hout, l.Wout, C, s));
  } else {
    {
      Prof p(c, PC_OTHER, 0, 0, s);
      MI355_TRY(launch_maxpool_bwd(c->dtype, G, c->pool_idx, B1, c->N, l.Hout, l.Wout, 64, s));
    }
    MI355_TRY(bn_backward(c, l, B1, c->a0_bits, nullptr, B1, beta_acc, s));
  }
  WgradArgs a;
  build_stem_wgrad_args(a, c->N, c->H, c->W);
  a.dy = B1; a.x = c->xpad; a.partial = c->wg_partial;
  hipStream_t ws;
  MI355_TRY(fork(c, s, &ws));
  {
    Prof p(c, PC_WGRAD64, conv_flops(c, l), 0, ws);
    MI355_TRY(launch_wgrad(c->dtype, a, l.splits, ws));
    snprintf(l.k_wgrad, sizeof(l.k_wgrad), "%s", mi355_last_conv_kernel());
  }
  MI355_TRY(launch_stem_unpack(c->wg_partial, l.splits, c->grads + l.w_off, beta_acc, ws));
  return release_set(c, par);
}

}  // namespace

extern "C" {

int mi355_resnet50_create(mi355_ctx** out, int device, int dtype, int N, int H, int W, int num_classes) {
  MI355_ARG(out, "create: null out");
  MI355_ARG(dtype == MI355_F32 || dtype == MI355_BF16 || dtype == MI355_FP8, "create: bad dtype %d", dtype);
  const bool fp8 = dtype == MI355_FP8;  // bf16 tensors + e4m3 conv operands
  if (fp8) dtype = MI355_BF16;
  MI355_ARG(N >= 1 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, "create: N=%d H=%d W=%d (H,W multiples of 32)",
            N, H, W);
  MI355_ARG(num_classes >= 1 && num_classes <= 65536, "create: num_classes=%d", num_classes);
  if (device >= 0) MI355_HIP(hipSetDevice(device));
  mi355_ctx* c = new mi355_ctx();
  c->device = device; c->dtype = dtype; c->N = N; c->H = H; c->W = W; c->num_classes = num_classes;
  c->fc_pad = (int)align_up((size_t)num_classes, 128);
  c->es = dtype_size(dtype);
  c->fp8 = fp8;

  // ---- network description ----
  init_conv(c->stem, "conv1", "bn1", 3, 64, 7, 2, H, W);
  c->stem.is_stem = true;
  int h = c->stem.Hout / 2, w = c->stem.Wout / 2, cin = 64;
  const int nblk[4] = {3, 4, 6, 3};
  const int planes[4] = {64, 128, 256, 512};
  for (int st = 0; st < 4; ++st) {
    for (int i = 0; i < nblk[st]; ++i) {
      Block b;
      const int p = planes[st];
      const int stride = (i == 0 && st > 0) ? 2 : 1;
      const std::string pre = "layer" + std::to_string(st + 1) + "." + std::to_string(i) + ".";
      b.Hin = h; b.Win = w; b.Cin = cin; b.Cout = 4 * p;
      init_conv(b.c1, pre + "conv1", pre + "bn1", cin, p, 1, 1, h, w);
      init_conv(b.c2, pre + "conv2", pre + "bn2", p, p, 3, stride, h, w);
      init_conv(b.c3, pre + "conv3", pre + "bn3", p, 4 * p, 1, 1, b.c2.Hout, b.c2.Wout);
      b.has_ds = (i == 0);
      if (b.has_ds) init_conv(b.ds, pre + "downsample.0", pre + "downsample.1", cin, 4 * p, 1, stride, h, w);
      b.Hout = b.c2.Hout; b.Wout = b.c2.Wout;
      h = b.Hout; w = b.Wout; cin = 4 * p;
      c->blocks.push_back(b);
    }
  }
  // ---- flat parameter layout, reverse execution order ----
  std::vector<TensorInfo> rev;
  c->fc_grad_begin = c->param_elems;
  add_param(c, rev, "fc.bias", &c->fc_b_off, 1, num_classes);
  add_param(c, rev, "fc.weight", &c->fc_w_off, 2, num_classes, 2048, 0, 0,
            (size_t)(c->fc_pad - num_classes) * 2048);
  c->fc_grad_end = c->param_elems;
  for (int i = (int)c->blocks.size() - 1; i >= 0; --i) {
    Block& b = c->blocks[i];
    b.grad_begin = c->param_elems;
    if (b.has_ds) register_convbn(c, rev, b.ds);
    register_convbn(c, rev, b.c3);
    register_convbn(c, rev, b.c2);
    register_convbn(c, rev, b.c1);
    b.grad_end = c->param_elems;
  }
  c->stem_grad_begin = c->param_elems;
  register_convbn(c, rev, c->stem);
  c->stem_grad_end = c->param_elems;
  // expose tensors in torchvision (forward) order
  c->tensors.assign(rev.rbegin(), rev.rend());

  // ---- FLOPs ----
  double f = conv_flops(c, c->stem), bw = 2 * conv_flops(c, c->stem) - conv_flops(c, c->stem);  // stem: no dgrad
  double tr = 2 * conv_flops(c, c->stem);
  (void)bw;
  for (auto& b : c->blocks) {
    const double fb = conv_flops(c, b.c1) + conv_flops(c, b.c2) + conv_flops(c, b.c3) + (b.has_ds ? conv_flops(c, b.ds) : 0);
    f += fb;
    tr += 3 * fb;
  }
  const double ffc = 2.0 * N * 2048 * num_classes;
  c->fwd_flops = f + ffc;
  c->train_flops = tr + 3 * ffc;

  // ---- workspace ----
  // executor switches: closed domains, read here once (a value outside its domain fails the creation below)
  const char* bad = nullptr;
  if (c->fp8) {
    plan_fp8(c, &bad);
    c->fp8_use_fwd = env_switch("MI355_FP8_FWD", 1, 1, &bad) != 0;
    c->fp8_use_bwd = env_switch("MI355_FP8_BWD", 1, 1, &bad) != 0;
  }
  c->stream_k = env_switch("MI355_STREAM_K", 1, 1, &bad) != 0;
  // BN-backward sums in the dgrad epilogues: measured same-box -0.6 ms/step in bf16, +0.9 ms in fp32 (the fp32 epilogue is already register-heavy)
  c->fuse_bn_bwd = env_switch("MI355_FUSE_BN_BWD", dtype == MI355_BF16 ? 1 : 0, 1, &bad) != 0;
  c->stem_fused_bwd = env_switch("MI355_STEM_FUSED", 1, 1, &bad) != 0;
  c->overlap = env_switch("MI355_WGRAD_STREAM", 1, 1, &bad) != 0;   // 0 keeps everything on the caller's stream
  c->ds_compact = env_switch("MI355_DS_COMPACT", 1, 1, &bad) != 0;
  if (bad) {
    set_error("create: %s=%s is outside the switch's domain", bad, getenv(bad));
    mi355_resnet50_destroy(c);
    return MI355_E_ARG;
  }
  Arena ar;
  plan_arena(c, ar);
  c->arena_bytes = ar.size;
  if (device < 0) {  // layout-only ctx: parameter table / sizes / FLOPs, no device memory
    *out = c;
    return 0;
  }
  hipError_t e = hipMalloc((void**)&c->arena, ar.size);
  if (e != hipSuccess) {
    set_error("create: hipMalloc(%zu bytes) -> %s", ar.size, hipGetErrorString(e));
    delete c;
    return MI355_E_NOMEM;
  }
  for (auto& sl : ar.slots) *sl.first = c->arena + sl.second;
  e = hipMemset(c->arena, 0, ar.size);
  if (e != hipSuccess) {
    set_error("create: hipMemset -> %s", hipGetErrorString(e));
    (void)hipFree(c->arena);
    delete c;
    return MI355_E_HIP;
  }
  if (c->fp8) {  // twins of the conv inputs (static graph) and unit scales until the first amaxes exist
    for (size_t i = 0; i < c->blocks.size(); ++i) {
      Block& b = c->blocks[i];
      if (i > 0) b.c1.in_q = b.ds.in_q = c->blocks[i - 1].out_q;
      b.c2.in_q = b.a1_q;
      b.c3.in_q = b.a2_q;
    }
    std::vector<float> ones(c->q_n, 1.f);
    if (hipMemcpy(c->q_scale, ones.data(), ones.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("create: fp8 scale table upload failed");
      mi355_resnet50_destroy(c);
      return MI355_E_HIP;
    }
  }
  // descriptor tables of the one-launch weight preparation
  {
    std::vector<PrepDesc> tab[2];
    int tiles = 0;
    auto add = [&](ConvBN& l) {
      PrepDesc d;
      d.w_off = l.w_off; d.w_cast = l.w_cast; d.w_tr = nullptr;
      d.w_q = d.w_trq = nullptr; d.q_scale = nullptr; d.q_amax = nullptr;
      d.Cout = l.Cout; d.taps = l.K * l.K; d.Cin = l.Cin; d.tile_begin = tiles;
      tab[0].push_back(d);
      d.w_tr = l.w_tr;
      if (l.qid_w >= 0) {
        d.w_q = l.w_q; d.w_trq = l.w_trq;
        d.q_scale = c->q_scale + l.qid_w; d.q_amax = c->q_amax + l.qid_w;
      }
      tab[1].push_back(d);
      tiles += (l.Cout / PREP_TILE) * (l.Cin / PREP_TILE) * l.K * l.K;
    };
    for (auto& b : c->blocks) {
      add(b.c1); add(b.c2); add(b.c3);
      if (b.has_ds) add(b.ds);
    }
    c->prep_layers = (int)tab[0].size();
    c->prep_tiles = tiles;
    bool ok = c->prep_layers <= 64;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipMemcpy(c->prep_table[i], tab[i].data(), tab[i].size() * sizeof(PrepDesc), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
      set_error("create: weight-prep table upload failed");
      mi355_resnet50_destroy(c);
      return MI355_E_HIP;
    }
  }
  // weight-gradient side stream (MI355_WGRAD_STREAM=0 keeps everything on the caller's stream)
  if (c->stream_k && dtype == MI355_F32) {
    if (hipHostMalloc((void**)&c->sk_err_host, 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
      set_error("create: hipHostMalloc -> %s", hipGetErrorString(hipGetLastError()));
      mi355_resnet50_destroy(c);
      return MI355_E_HIP;
    }
    c->sk_err_host[0] = c->sk_err_host[1] = 0;
  }
  if (c->overlap) {
    // the weight-gradient stream runs at the highest priority: it is the busier of the two during backward and the main stream ends
    // up waiting for it (profiles/r04_ab_side_stream_priority.txt: -0.07..-0.13 ms per step on three boxes)
    bool ok;
    {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      ok = hipStreamCreateWithPriority(&c->wstream, hipStreamNonBlocking, hi) == hipSuccess;
    }
    c->fork_ev.resize(16);
    for (auto& ev : c->fork_ev) ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    for (auto& ev : c->w_done) ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ds_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      set_error("create: side stream / events -> %s", hipGetErrorString(hipGetLastError()));
      mi355_resnet50_destroy(c);
      return MI355_E_HIP;
    }
  }
  // block inputs
  const void* prev = c->p0;
  for (auto& b : c->blocks) {
    b.in = prev;
    prev = b.out;
  }
  *out = c;
  return 0;
}

int mi355_resnet50_destroy(mi355_ctx* c) {
  if (!c) return 0;
  if (c->device >= 0) (void)hipSetDevice(c->device);
  for (auto e : c->ev) (void)hipEventDestroy(e);
  for (auto e : c->fork_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto e : c->w_done)
    if (e) (void)hipEventDestroy(e);
  if (c->ds_done) (void)hipEventDestroy(c->ds_done);
  if (c->wstream) (void)hipStreamDestroy(c->wstream);
  if (c->sk_err_host) (void)hipHostFree(c->sk_err_host);
  if (c->arena) (void)hipFree(c->arena);
  delete c;
  return 0;
}

int mi355_resnet50_num_tensors(const mi355_ctx* c) { return c ? (int)c->tensors.size() : 0; }

int mi355_resnet50_tensor_info(const mi355_ctx* c, int idx, char* name, int name_cap, int* kind, size_t* offset,
                               int* ndim, int shape[4]) {
  MI355_ARG(c && idx >= 0 && idx < (int)c->tensors.size(), "tensor_info: bad index %d", idx);
  const TensorInfo& t = c->tensors[idx];
  if (name && name_cap > 0) {
    strncpy(name, t.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  if (ndim) *ndim = t.ndim;
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  return 0;
}

size_t mi355_resnet50_flat_param_elems(const mi355_ctx* c) { return c ? c->param_elems : 0; }
size_t mi355_resnet50_flat_buffer_elems(const mi355_ctx* c) { return c ? c->buffer_elems : 0; }
size_t mi355_resnet50_workspace_bytes(const mi355_ctx* c) { return c ? c->arena_bytes : 0; }

int mi355_resnet50_bind(mi355_ctx* c, float* params, float* grads, float* buffers) {
  MI355_ARG(c && params && buffers, "bind: null pointer");
  if (c->device < 0) {
    set_error("bind: layout-only ctx (created with device < 0)");
    return MI355_E_STATE;
  }
  MI355_ARG(((uintptr_t)params % 256 == 0) && ((uintptr_t)buffers % 256 == 0) && ((uintptr_t)grads % 256 == 0),
            "bind: flat arrays must be 256-byte aligned");
  c->params = params; c->grads = grads; c->buffers = buffers;
  return 0;
}

int mi355_resnet50_forward(mi355_ctx* c, const float* x_nchw, float* logits, int training, float bn_momentum,
                           void* stream) {
  MI355_ARG(c && x_nchw && logits, "forward: null pointer");
  if (!c->params) {
    set_error("forward: parameters not bound");
    return MI355_E_STATE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int N = c->N;
  MI355_TRY(sk_check(c));
  c->fwd_training_done = false;
  for (auto& b : c->blocks) b.c1.bwd_rows = b.c2.bwd_rows = b.c3.bwd_rows = b.ds.bwd_rows = 0;
  if (c->fp8 && training) {
    // delayed scaling: the amaxes the previous step's producers recorded become this step's scales; the very first training
    // step of a ctx has none yet and runs its convs on the bf16 tensors (the twins are still written: that records the amaxes)
    c->fp8_fwd_on = c->fp8_fwd_cal && c->fp8_use_fwd;
    c->fp8_bwd_on = c->fp8_fwd_cal && c->fp8_bwd_cal && c->fp8_use_bwd;
    c->fp8_lean = c->fp8_fwd_on && c->fp8_bwd_on && !c->fp8_keep_bf16;
    if (c->fp8_fwd_cal) MI355_TRY(launch_fp8_scale_update(c->q_scale, c->q_amax, c->q_n, FP8_HEADROOM, s));
  }
  MI355_TRY(weight_prep_all(c, training != 0, s));
  {
    Prof p(c, PC_OTHER, 0, 0, s);
    MI355_TRY(launch_stem_ingest(c->dtype, x_nchw, c->xpad, N, c->H, c->W, s));
  }
  // stem
  {
    IgemmArgs a;
    build_stem_fwd_args(a, N, c->H, c->W);
    a.in = c->xpad; a.wt = c->stem_pack; a.out = c->stem.y;
    a.stat_partial = training ? c->bn_partial : nullptr;
    Prof p(c, PC_IGEMM64, conv_flops(c, c->stem), 0, s);
    MI355_TRY(launch_igemm(c->dtype, a, 1, s, &c->stem.stat_rows));
    snprintf(c->stem.k_fwd, sizeof(c->stem.k_fwd), "%s", mi355_last_conv_kernel());
  }
  MI355_TRY(bn_prepare(c, c->stem, training, bn_momentum, s));
  {
    // BN + ReLU + 3x3/2 max pool in one pass over the raw stem output; the full-resolution activation is never stored
    const int C = 64;
    Prof p(c, PC_BN_APPLY, 0, (double)N * c->stem.Hout * c->stem.Wout * C * c->es * 1.25, s);
    MI355_TRY(launch_bn_relu_maxpool(c->dtype, c->stem.y, c->stem.stat + 2 * C, c->stem.stat + 3 * C, c->p0, c->pool_idx,
                                     c->a0_bits, N, c->stem.Hout, c->stem.Wout, C, s));
  }
  for (auto& b : c->blocks) {
    if (b.has_ds) {  // the downsample conv + its statistics run beside conv1..conv3
      hipStream_t ws;
      MI355_TRY(fork(c, s, &ws));
      MI355_TRY(conv_forward(c, b.ds, b.in, training, bn_momentum, ws));
      MI355_TRY(bn_prepare(c, b.ds, training, bn_momentum, ws));
      if (c->overlap) MI355_HIP(hipEventRecord(c->ds_done, ws));
    }
    MI355_TRY(conv_forward(c, b.c1, b.in, training, bn_momentum, s));
    MI355_TRY(bn_prepare(c, b.c1, training, bn_momentum, s));
    // (fp8 step, lean: a1 / a2 are read by conv2 / conv3 forward and their weight gradients only — all through the twin)
    if (conv_bn_in_legal(c, b.c1, b.c2, training)) {
      // bn1 + ReLU in conv2's operand path: a1 and its bits are by-products of that launch (no bn_apply pass over y1)
      MI355_TRY(conv_forward(c, b.c2, b.c1.y, training, bn_momentum, s, &b.c1, b.a1, b.a1_bits));
    } else {
      MI355_TRY(bn_apply(c, b.c1, nullptr, nullptr, b.a1, 1, s, training ? b.a1_bits : nullptr, b.a1_q, b.qid_a1,
                         training && c->fp8_lean && b.c2.fp8_fwd && b.c2.fp8_wgrad));
      MI355_TRY(conv_forward(c, b.c2, b.a1, training, bn_momentum, s));
    }
    MI355_TRY(bn_prepare(c, b.c2, training, bn_momentum, s));
    if (conv_bn_in_legal(c, b.c2, b.c3, training)) {
      MI355_TRY(conv_forward(c, b.c3, b.c2.y, training, bn_momentum, s, &b.c2, b.a2, b.a2_bits));   // bn2 + ReLU in conv3's operand path
    } else {
      MI355_TRY(bn_apply(c, b.c2, nullptr, nullptr, b.a2, 1, s, training ? b.a2_bits : nullptr, b.a2_q, b.qid_a2,
                         training && c->fp8_lean && b.c3.fp8_fwd && b.c3.fp8_wgrad));
      MI355_TRY(conv_forward(c, b.c3, b.a2, training, bn_momentum, s));
    }
    MI355_TRY(bn_prepare(c, b.c3, training, bn_momentum, s));
    if (b.has_ds) {
      if (c->overlap) MI355_HIP(hipStreamWaitEvent(s, c->ds_done, 0));
      MI355_TRY(bn_apply(c, b.c3, nullptr, &b.ds, b.out, 1, s, training ? b.out_bits : nullptr, b.out_q, b.qid_out));
    } else {
      MI355_TRY(bn_apply(c, b.c3, b.in, nullptr, b.out, 1, s, training ? b.out_bits : nullptr, b.out_q, b.qid_out));
    }
  }
  const Block& last = c->blocks.back();
  {
    Prof p(c, PC_OTHER, 0, 0, s);
    MI355_TRY(launch_gap_fwd(c->dtype, last.out, c->pooled, N, last.Hout * last.Wout, last.Cout, s));
    MI355_TRY(launch_fc(c->pooled, c->params + c->fc_w_off, c->fc_tmp, N, c->fc_pad, 2048, s));
    MI355_TRY(launch_bias_slice(c->fc_tmp, c->fc_pad, c->params + c->fc_b_off, logits, N, c->num_classes, s));
  }
  c->fwd_training_done = training != 0;
  if (training) c->fp8_fwd_cal = true;
  c->next_seg = 0;
  return sk_snapshot(c, s);
}

int mi355_resnet50_num_segments(const mi355_ctx* c) { return c ? (int)c->blocks.size() + 2 : 0; }

int mi355_resnet50_segment_range(const mi355_ctx* c, int seg, size_t* gb, size_t* ge) {
  MI355_ARG(c && seg >= 0 && seg < (int)c->blocks.size() + 2, "segment_range: bad segment %d", seg);
  size_t b, e;
  seg_range(c, seg, &b, &e);
  if (gb) *gb = b;
  if (ge) *ge = e;
  return 0;
}

int mi355_resnet50_bucket_plan(const mi355_ctx* c, double bucket_cap_mb, int cap, int* n_out, size_t* begins, size_t* ends,
                               int* last_segs) {
  MI355_ARG(c && n_out && bucket_cap_mb > 0, "bucket_plan: bad arguments");
  const auto bk = plan_buckets(c, (size_t)(bucket_cap_mb * (1 << 20) / 4));
  *n_out = (int)bk.size();
  for (int i = 0; i < (int)bk.size() && i < cap; ++i) {
    if (begins) begins[i] = bk[i].begin;
    if (ends) ends[i] = bk[i].end;
    if (last_segs) last_segs[i] = bk[i].last_seg;
  }
  return 0;
}

int mi355_resnet50_set_comm(mi355_ctx* c, mi355_comm* comm, double bucket_cap_mb) {
  MI355_ARG(c && (comm == nullptr || bucket_cap_mb > 0), "set_comm: bad arguments");
  if (c->device < 0) {
    set_error("set_comm: layout-only ctx (created with device < 0)");
    return MI355_E_STATE;
  }
  c->comm = comm;
  c->buckets = comm ? plan_buckets(c, (size_t)(bucket_cap_mb * (1 << 20) / 4)) : std::vector<mi355_ctx::Bucket>();
  return 0;
}

int mi355_resnet50_set_grad_sync(mi355_ctx* c, int on) {
  MI355_ARG(c, "set_grad_sync: null ctx");
  c->grad_sync = on != 0;
  return 0;
}

int mi355_resnet50_backward(mi355_ctx* c, const float* dlogits, int seg_begin, int seg_end, int accumulate,
                            void* stream) {
  MI355_ARG(c, "backward: null ctx");
  const int nseg = (int)c->blocks.size() + 2;
  MI355_ARG(seg_begin >= 0 && seg_end <= nseg && seg_begin < seg_end, "backward: bad segment range [%d,%d)", seg_begin,
            seg_end);
  if (!c->fwd_training_done || !c->grads) {
    set_error("backward: needs a training forward and bound gradients first");
    return MI355_E_STATE;
  }
  if (seg_begin != c->next_seg) {
    set_error("backward: segments must run in order (expected %d, got %d)", c->next_seg, seg_begin);
    return MI355_E_STATE;
  }
  hipStream_t s = (hipStream_t)stream;
  MI355_TRY(sk_check(c));
  const float beta_acc = accumulate ? 1.f : 0.f;
  const int nb = (int)c->blocks.size();
  for (int seg = seg_begin; seg < seg_end; ++seg) {
    if (seg == 0) {
      MI355_ARG(dlogits, "backward: dlogits is null");
      MI355_TRY(backward_fc(c, dlogits, beta_acc, s));
    } else if (seg == nb + 1) {
      MI355_TRY(backward_stem(c, beta_acc, s));
    } else {
      MI355_TRY(backward_block(c, c->blocks[nb - seg], nb - seg > 0 ? &c->blocks[nb - seg - 1] : nullptr, beta_acc, s));
    }
    c->next_seg = seg + 1;
    if (c->comm && c->grad_sync)
      for (const auto& bk : c->buckets)
        if (bk.last_seg == seg) {  // the bucket's producers are all enqueued (main + weight-gradient stream): reduce it
          MI355_TRY(comm_allreduce_bucket(c->comm, c->grads, bk.begin, bk.end, s, c->overlap && c->w_dirty ? c->wstream : nullptr));
          c->comm_dirty = true;
        }
  }
  MI355_TRY(join(c, s));
  if (c->comm && c->comm_dirty) {
    MI355_TRY(comm_join(c->comm, s));
    c->comm_dirty = false;
  }
  if (c->next_seg == nseg) {
    c->fwd_training_done = false;
    c->fp8_bwd_cal = true;
  }
  return sk_snapshot(c, s);
}

int mi355_resnet50_debug_tensor(const mi355_ctx* c, const char* name, void** ptr, int* dtype, int* ndim,
                                int shape[4]) {
  MI355_ARG(c && name && ptr && dtype && ndim && shape, "debug_tensor: null argument");
  const std::string n(name);
  auto set4 = [&](void* p, int dt, int h, int w, int ch) {
    *ptr = p; *dtype = dt; *ndim = 4;
    shape[0] = c->N; shape[1] = h; shape[2] = w; shape[3] = ch;
    return 0;
  };
  auto set1 = [&](void* p, int len) {
    *ptr = p; *dtype = MI355_F32; *ndim = 1;
    shape[0] = len; shape[1] = shape[2] = shape[3] = 0;
    return 0;
  };
  auto setq = [&](const void* p, int d0, int d1, int d2, int d3) {  // e4m3 bytes
    if (!p) return 1;
    *ptr = const_cast<void*>(p); *dtype = MI355_FP8; *ndim = 4;
    shape[0] = d0; shape[1] = d1; shape[2] = d2; shape[3] = d3;
    return 0;
  };
  auto conv_match = [&](const ConvBN& l) -> int {
    if (c->fp8 && l.qid_w >= 0) {  // fp8 step: the operands the layer's convs read and their scales of the last step
      if (n == l.conv_name + ".xq" && l.fp8_fwd) return setq(l.in_q, c->N, l.Hin, l.Win, l.Cin);
      if (n == l.conv_name + ".wq") return setq(l.w_q, l.Cout, l.K, l.K, l.Cin);
      if (n == l.conv_name + ".wtq") return setq(l.w_trq, l.Cin, l.K, l.K, l.Cout);
      if (n == l.conv_name + ".sx" && l.fp8_fwd) return set1(c->q_scale + l.qid_in, 1);
      if (n == l.conv_name + ".sw") return set1(c->q_scale + l.qid_w, 1);
      if (n == l.conv_name + ".sdy" && l.fp8_dgrad) return set1(c->q_scale + l.qid_dy, 1);
      if (n == l.conv_name + ".amax_w") return set1(reinterpret_cast<float*>(c->q_amax + l.qid_w), 1);
    }
    if (n == l.conv_name + ".y") return set4(l.y, c->dtype, l.Hout, l.Wout, l.Cout);
    if (n == l.bn_name + ".save_mean") return set1(l.stat, l.Cout);
    if (n == l.bn_name + ".save_invstd") return set1(l.stat + l.Cout, l.Cout);
    return 1;
  };
  if (conv_match(c->stem) == 0) return 0;
  if (n == "stem.p0") return set4(c->p0, c->dtype, c->stem.Hout / 2, c->stem.Wout / 2, 64);
  for (const auto& b : c->blocks) {
    if (conv_match(b.c1) == 0 || conv_match(b.c2) == 0 || conv_match(b.c3) == 0) return 0;
    if (b.has_ds && conv_match(b.ds) == 0) return 0;
    const std::string pre = b.c1.conv_name.substr(0, b.c1.conv_name.size() - 5);  // strip "conv1"
    if (n == pre + "a1") return set4(b.a1, c->dtype, b.c1.Hout, b.c1.Wout, b.c1.Cout);
    if (n == pre + "a2") return set4(b.a2, c->dtype, b.c2.Hout, b.c2.Wout, b.c2.Cout);
    if (n == pre + "out") return set4(b.out, c->dtype, b.Hout, b.Wout, b.Cout);
  }
  const Block& last = c->blocks.back();
  if (n == "pooled" || n == "dpooled") {
    *ptr = n == "pooled" ? c->pooled : c->dpooled; *dtype = MI355_F32; *ndim = 2;
    shape[0] = c->N; shape[1] = 2048; shape[2] = shape[3] = 0;
    return 0;
  }
  for (int i = 0; i < 2; ++i)
    if (n == "gG" + std::to_string(i)) return set4(c->gG[i], c->dtype, last.Hout, last.Wout, last.Cout);
  if (n == "grad.cur") {  // between two backward segments: the gradient the NEXT segment starts from (wrt that block's output, before its ReLU mask)
    const int nb = (int)c->blocks.size();
    if (c->next_seg >= 1 && c->next_seg <= nb) {
      const Block& b = c->blocks[nb - c->next_seg];
      return set4(c->cur_dout, c->dtype, b.Hout, b.Wout, b.Cout);
    }
    if (c->next_seg == nb + 1) return set4(c->cur_dout, c->dtype, c->stem.Hout / 2, c->stem.Wout / 2, 64);
    set_error("debug_tensor: 'grad.cur' exists between backward segments only (next segment %d)", c->next_seg);
    return MI355_E_STATE;
  }
  set_error("debug_tensor: unknown tensor '%s'", name);
  return MI355_E_ARG;
}

int mi355_resnet50_force_grad(mi355_ctx* c, const void* g, size_t bytes, void* stream) {
  MI355_ARG(c && g, "force_grad: null argument");
  const int nb = (int)c->blocks.size();
  if (!c->fwd_training_done || c->next_seg < 1 || c->next_seg > nb + 1) {
    set_error("force_grad: only between two backward segments (next segment %d)", c->next_seg);
    return MI355_E_STATE;
  }
  size_t want;
  if (c->next_seg <= nb) {
    Block& b = c->blocks[nb - c->next_seg];
    want = (size_t)c->N * b.Hout * b.Wout * b.Cout * c->es;
    b.c3.bwd_rows = 0;  // the BN-backward sums the producing epilogue left belong to the replaced tensor: bn3's backward re-reduces
  } else {
    want = (size_t)c->N * (c->stem.Hout / 2) * (c->stem.Wout / 2) * 64 * c->es;
  }
  MI355_ARG(bytes == want, "force_grad: %zu bytes given, the pending gradient has %zu", bytes, want);
  MI355_HIP(hipMemcpyAsync(c->cur_dout, g, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int mi355_resnet50_fp8_state(const mi355_ctx* c, int* fwd_on, int* bwd_on, int* n_fwd_layers, int* n_dgrad_layers) {
  MI355_ARG(c, "fp8_state: null ctx");
  int nf = 0, nd = 0;
  auto count = [&](const ConvBN& l) {
    nf += l.fp8_fwd ? 1 : 0;
    nd += l.fp8_dgrad ? 1 : 0;
  };
  for (const auto& b : c->blocks) {
    count(b.c1); count(b.c2); count(b.c3);
    if (b.has_ds) count(b.ds);
  }
  if (fwd_on) *fwd_on = c->fp8 && c->fp8_fwd_on;
  if (bwd_on) *bwd_on = c->fp8 && c->fp8_bwd_on;
  if (n_fwd_layers) *n_fwd_layers = nf;
  if (n_dgrad_layers) *n_dgrad_layers = nd;
  return 0;
}

int mi355_resnet50_flops(const mi355_ctx* c, double* fwd, double* train) {
  MI355_ARG(c, "flops: null ctx");
  if (fwd) *fwd = c->fwd_flops;
  if (train) *train = c->train_flops;
  return 0;
}

int mi355_resnet50_kernel_table(const mi355_ctx* c, char* out, size_t cap, size_t* needed) {
  MI355_ARG(c && (out || cap == 0), "kernel_table: bad arguments");
  std::string t;
  auto line = [&](const ConvBN& l) {
    t += l.conv_name + " fwd=" + (l.k_fwd[0] ? l.k_fwd : "-") + " dgrad=" + (l.k_dgrad[0] ? l.k_dgrad : "-") + " wgrad=" + (l.k_wgrad[0] ? l.k_wgrad : "-") + "\n";
  };
  line(c->stem);
  for (const Block& b : c->blocks) {
    line(b.c1); line(b.c2); line(b.c3);
    if (b.has_ds) line(b.ds);
  }
  if (needed) *needed = t.size() + 1;
  if (cap > 0) {
    const size_t n = std::min(cap - 1, t.size());
    memcpy(out, t.data(), n);
    out[n] = 0;
  }
  return 0;
}

int mi355_resnet50_profile(mi355_ctx* c, int class_mask) {
  MI355_ARG(c, "profile: null ctx");
  MI355_HIP(hipSetDevice(c->device));
  if (class_mask && c->ev.empty()) {
    c->ev.resize(MAX_EVENTS);
    for (auto& e : c->ev) MI355_HIP(hipEventCreate(&e));
  }
  c->prof_mask = (unsigned)class_mask;
  c->recs.clear();
  return 0;
}

int mi355_resnet50_profile_read(mi355_ctx* c, int kind, double* total_ms, int* launches, double* alg_flops,
                                double* alg_bytes) {
  MI355_ARG(c && kind >= 0 && kind < 32, "profile_read: bad arguments");
  double ms = 0, fl = 0, by = 0;
  int n = 0;
  for (size_t i = 0; i < c->recs.size(); ++i) {
    const bool match = kind == 8 ? ((c->recs[i].tag & 1) && c->recs[i].cls <= 1) : kind == 9 ? (c->recs[i].tag & 2) != 0 : c->recs[i].cls == kind;
    if (!match) continue;
    float t = 0;
    MI355_HIP(hipEventSynchronize(c->ev[2 * i + 1]));
    MI355_HIP(hipEventElapsedTime(&t, c->ev[2 * i], c->ev[2 * i + 1]));
    ms += t; fl += c->recs[i].flops; by += c->recs[i].bytes;
    ++n;
  }
  MI355_TRY(sk_check(c));
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (alg_flops) *alg_flops = fl;
  if (alg_bytes) *alg_bytes = by;
  return 0;
}

}  // extern "C"
