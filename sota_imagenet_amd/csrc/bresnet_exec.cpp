// bresnet_exec.cpp — BResNet-50 (BASELINE.json configs[3]) as a static executor: ONE C-ABI call per forward and per backward.
//
// The model is what the reference builds with `_target_: pytorch_tools.models.resnet50` and the model_params of
// configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51 (stem_type deep, antialias, attn_type eca, norm_layer
// inplaceabn, norm_act leaky_relu, drop_rate / drop_connect_rate 0.2) with every conv under weight standardisation
// (train.py:66-67, yaml:59).  pytorch_tools is not vendored; the block structure is the one SURVEY.md Appendix C recalls and
// oracle/bresnet50_ref.py restates:
//   deep stem   conv3x3(3,32,s2)+ABN, conv3x3(32,32)+ABN, conv3x3(32,64), bn1 = ABN(64); pool = maxpool 3x3/1 + BlurPool
//   bottleneck  conv1x1+ABN, conv3x3 (stride 1)+ABN [+BlurPool when the block strides], conv1x1+ABN(identity), ECA(k=3),
//               drop-connect on the branch, + shortcut ([AvgPool 2x2] conv1x1 ABN(identity) where the block strides / widens),
//               leaky ReLU
//   head        GAP, dropout, FC
// Round 2 drove this graph from Python, one C-ABI call per op through torch.autograd nodes, with torch cat / permute / .to() /
// foreach-SGD glue (13.8 % of the kernel time was at::native, profiles/r02d_bresnet50_kernel_stats.csv).  Here the same
// operator sequence — the same kernels, called through the same per-op entry points of include/mi355rn.h, in the same order, so
// the two agree bit for bit up to the FC GEMM — runs from C++ over
//   * ONE flat fp32 parameter array / gradient array / buffer array (what the native SGD and the bucketed all-reduce take), conv
//     weights in [Cout][KH][KW][Cin] order = torch OIHW in channels_last memory;
//   * a workspace laid out once per (N, H, W): every activation and every gradient has its own slot (50 GB in all at batch 256 /
//     224 px in bf16, of 288), so no launch waits for a buffer;
//   * weight standardisation + channel padding (the 3- and 32-channel stem tensors run zero-padded to 64) + cast in the weight
//     preparation at the head of each forward, the inverse (un-pad, standardisation backward) behind each weight gradient;
//   * the weight-gradient convolutions on a side stream (they only feed the optimizer), joined at the end of backward.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi355rn.h"
#include "comm.h"
#include "common.h"

namespace mi355 {
namespace {

constexpr int ACT_NONE = 0, ACT_LEAKY = 2;
constexpr float BN_EPS = 1e-5f, WS_EPS = 1e-5f;

struct TInfo {
  std::string name;
  int kind;  // 0 parameter, 1 buffer
  size_t off;
  int ndim;
  int shape[4];
};

struct VConv {
  std::string name;
  int Cin = 0, Cout = 0, K = 1, stride = 1, pad = 0, Hin = 0, Win = 0, Hout = 0, Wout = 0, Cinp = 0, Coutp = 0;
  size_t w_off = 0;
  float* w_hat = nullptr;   // standardised weights [Cout][K][K][Cin] (weight standardisation on)
  float* ws_mean = nullptr; // [Cout]
  float* ws_invstd = nullptr;
  void* wp = nullptr;       // [Coutp][K][K][Cinp] in the compute dtype: what the kernels read
  void* wtr = nullptr;      // [Cinp][K][K][Coutp]: the same weights transposed, what the data-gradient convolution reads
  void* y = nullptr;        // conv output [N][Hout][Wout][Coutp]
  void* dy = nullptr;       // its gradient
  bool padded() const { return Cin != Cinp || Cout != Coutp; }
  size_t welems() const { return (size_t)Cout * K * K * Cin; }
  size_t wpelems() const { return (size_t)Coutp * K * K * Cinp; }
};

struct VBN {
  std::string name;
  int C = 0, Cp = 0, act = 0;
  size_t g_off = 0, b_off = 0, rm_off = 0, rv_off = 0;
  float *stage = nullptr;   // Cp != C: [6][Cp] gamma, beta, running_mean, running_var, dgamma, dbeta (zero / one padded)
  float *mean = nullptr, *invstd = nullptr;  // [Cp] batch statistics kept for backward
  void* out = nullptr;      // activation
  void* dout = nullptr;     // gradient wrt the activation
  float *scale = nullptr, *shift = nullptr;  // [Cp] the normalisation as y * scale + shift (kept when the consumer applies it on the fly)
  bool lazy = false;        // last forward: `out` was NOT written, the consumer reads (conv output, scale, shift)
  uint8_t* bits = nullptr;  // leaky-ReLU mask of `out` (bit e of byte i = element e of 16-byte vector i was > 0), written by the training forward
  bool has_bits = false;    // ... by the last one
};

struct VBlock {
  int cin = 0, planes = 0, stride = 1, H = 0, W = 0, Ho = 0, Wo = 0;  // H, W: input resolution; Ho, Wo: output
  bool has_ds = false;
  VConv c1, c2, c3, ds;
  VBN b1, b2, b3, bd;
  size_t eca_off = 0;             // 3 conv1d weights
  void *a2b = nullptr, *da2b = nullptr;  // blur-pooled a2 (stride-2 blocks)
  void *scin = nullptr, *dscin = nullptr;  // avg-pooled block input (stride-2 blocks)
  void *e = nullptr, *de = nullptr;      // ECA output
  float *pooled = nullptr, *gate = nullptr, *keep = nullptr;
  void *out = nullptr;            // block output
  uint8_t* out_bits = nullptr;    // its sign bits (one byte per 16-byte vector), written by the fused ECA forward for the backward's activation slope
  bool has_out_bits = false;      // ... by the last forward
  void *dsc = nullptr;            // gradient wrt the shortcut operand
  void *dxb = nullptr, *dxs = nullptr, *dx = nullptr;  // gradient wrt the block input: branch, shortcut, sum
  bool scaled = false;            // the last forward scaled the branch by `keep` (drop-connect)
};

struct Arena {
  size_t size = 0;
  std::vector<std::pair<void**, size_t>> slots;
  template <typename P>
  void add(P** p, size_t bytes) {
    slots.push_back({reinterpret_cast<void**>(p), size});
    size += align_up(bytes, 256);
  }
};

}  // namespace
}  // namespace mi355

using namespace mi355;

struct mi355_bctx {
  int device = 0, dtype = 0, N = 0, H = 0, W = 0, num_classes = 0, fc_pad = 0;
  size_t es = 4;
  bool wstd = false;
  float drop_rate = 0.f, drop_connect = 0.f;
  unsigned long long seed = 0;
  std::vector<TInfo> tensors;
  size_t param_elems = 0, buffer_elems = 0;
  float *params = nullptr, *grads = nullptr, *buffers = nullptr;
  VConv s0, s1, s2;
  VBN sb0, sb1, sb2;
  std::vector<VBlock> blocks;
  size_t fc_w_off = 0, fc_b_off = 0;
  // workspace
  char* arena = nullptr;
  size_t arena_bytes = 0;
  void *h0 = nullptr, *m = nullptr, *dm = nullptr, *p = nullptr;
  uint8_t* pool_idx = nullptr;
  float *pooled = nullptr, *pooled_d = nullptr, *do_mask = nullptr, *fc_tmp = nullptr, *dlogits_pad = nullptr, *dpooled = nullptr, *fc_wtr = nullptr;
  void* dlast = nullptr;  // gradient wrt the last block's output
  float* partial = nullptr;
  size_t partial_bytes = 0;
  void* bn_ws = nullptr;
  size_t bn_ws_bytes = 0;
  void* wg_ws = nullptr;   // split-K slabs of the running wgrad (side stream)
  size_t wg_ws_bytes = 0;
  float* dw_tmp = nullptr; // padded / pre-standardisation weight gradient (side stream)
  float* eca_ws = nullptr;
  // streams
  hipStream_t wstream = nullptr;
  std::vector<hipEvent_t> ev;
  size_t ev_next = 0;
  bool overlap = true, w_dirty = false;
  // gradient collective inside the boundary (comm.cpp), as in resnet_exec.cpp: backward segments (0 = head, then the bottlenecks last
  // to first, then the stem; the flat array is laid out in FORWARD order, so segments descend through it) form buckets, each reduced by
  // one mean all-reduce on the communicator's stream as soon as its last segment has been enqueued on both streams
  struct Bucket { size_t begin, end; int last_seg; };
  std::vector<std::pair<size_t, size_t>> segs;
  std::vector<Bucket> buckets;
  mi355_comm* comm = nullptr;
  bool grad_sync = true, comm_dirty = false;
  // test hooks of the NEXT backward call (mi355_bresnet50_grad_hooks): per block, where to copy the gradient wrt its output before its backward runs,
  // and what to overwrite it with (teacher forcing between backward segments)
  std::vector<void*> hook_record;
  std::vector<const void*> hook_replay;
  bool batch_prep = true;  // every convolution's weight preparation in two launches (MI355_BRESNET_BATCH_PREP=0: three small launches per layer)
  BPrepDesc* prep_table = nullptr;       // device copy of the table (built for the bound parameter array)
  const float* prep_params = nullptr;    // ... the array it was built for
  int prep_n = 0, prep_rows = 0, prep_tiles = 0;
  bool fuse_bn_bwd = true; // bn1's / bn2's (and the deep stem's) backward sums in the epilogue of the data gradient in front of them (MI355_BRESNET_FUSE_BN_BWD=0: reduction passes)
  bool fused_add = true;   // shortcut gradient added in conv1's dgrad epilogue (MI355_BRESNET_FUSED_ADD=0: its own launch, as the per-op graph)
  const char* bad_switch = nullptr;   // an environment switch with a value outside its domain (the creation fails)
  bool use_bits = true;    // leaky-ReLU masks travel as one bit per element (round 3: 36.5 -> 35.5 ms; the A/B switch is gone)
  bool lazy_dz3 = true;    // bn3's backward forms its input gradient from the ECA backward on the fly (round 3: 34.5 -> 33.3 ms; the A/B switch is gone)
  bool lazy_bn = true;     // bn3 / downsample BN normalised inside the fused ECA pass, their outputs never stored (MI355_BRESNET_LAZY_BN=0: stored)
  bool stem_im2col = true; // the first stem convolution (3 -> 32, 3x3 / 2) as a 1x1 convolution over its 27-value patches (MI355_BRESNET_STEM_IM2COL=0: 3x3 over the 64-channel input)
  bool eca_sums = true;    // bn3's BatchNorm-backward sums from the per-image sums of the fused ECA backward: no reduction pass over the tensors (MI355_BRESNET_ECA_SUMS=0: its own pass)
  bool fused_eca = true;   // ECA gate x drop-connect x shortcut add x activation in one pass each way (MI355_BRESNET_FUSED_ECA=0: op by op)
  bool have_fwd = false, dropped = false;  // state of the last forward: training pass / dropout mask in use
  double fwd_flops = 0;
};

namespace {

size_t reg(mi355_bctx* c, const std::string& name, int kind, std::initializer_list<int> shape) {
  TInfo t;
  t.name = name;
  t.kind = kind;
  size_t n = 1;
  t.ndim = 0;
  for (int s : shape) {
    t.shape[t.ndim++] = s;
    n *= (size_t)s;
  }
  size_t& cur = kind == 0 ? c->param_elems : c->buffer_elems;
  cur = align_up(cur, 64);  // 256-byte aligned tensors: the kernels load per-channel constants as 16-byte vectors
  t.off = cur;
  cur += n;
  c->tensors.push_back(t);
  return t.off;
}

int up64(int v) { return (v + 63) / 64 * 64; }

void init_conv(mi355_bctx* c, VConv& v, const std::string& name, int Cin, int Cout, int K, int stride, int H, int W) {
  v.name = name;
  v.Cin = Cin; v.Cout = Cout; v.K = K; v.stride = stride; v.pad = K / 2;
  v.Hin = H; v.Win = W;
  v.Hout = (H + 2 * v.pad - K) / stride + 1;
  v.Wout = (W + 2 * v.pad - K) / stride + 1;
  v.Cinp = up64(Cin); v.Coutp = up64(Cout);
  v.w_off = reg(c, name + ".weight", 0, {Cout, Cin, K, K});
  c->fwd_flops += 2.0 * c->N * v.Hout * v.Wout * (double)Cout * Cin * K * K;
}

void init_bn(mi355_bctx* c, VBN& b, const std::string& name, int C, int act) {
  b.name = name;
  b.C = C; b.Cp = up64(C); b.act = act;
  b.g_off = reg(c, name + ".weight", 0, {C});
  b.b_off = reg(c, name + ".bias", 0, {C});
  b.rm_off = reg(c, name + ".running_mean", 1, {C});
  b.rv_off = reg(c, name + ".running_var", 1, {C});
}

void plan_conv(mi355_bctx* c, Arena& ar, VConv& v) {
  if (c->wstd) {
    ar.add(&v.w_hat, v.welems() * 4);
    ar.add(&v.ws_mean, (size_t)v.Cout * 4);
    ar.add(&v.ws_invstd, (size_t)v.Cout * 4);
  }
  ar.add(&v.wp, v.wpelems() * c->es);
  ar.add(&v.wtr, v.wpelems() * c->es);
  const size_t o = (size_t)c->N * v.Hout * v.Wout * v.Coutp * c->es;
  ar.add(&v.y, o);
  ar.add(&v.dy, o);
}
void plan_bn(mi355_bctx* c, Arena& ar, VBN& b, int H, int W, bool want_dout = true) {
  if (b.Cp != b.C) ar.add(&b.stage, (size_t)6 * b.Cp * 4);
  ar.add(&b.mean, (size_t)b.Cp * 4);
  ar.add(&b.invstd, (size_t)b.Cp * 4);
  ar.add(&b.scale, (size_t)b.Cp * 4);
  ar.add(&b.shift, (size_t)b.Cp * 4);
  const size_t o = (size_t)c->N * H * W * b.Cp * c->es;
  ar.add(&b.out, o);
  if (want_dout) ar.add(&b.dout, o);
  if (b.act != ACT_NONE) ar.add(&b.bits, o / 16);
}

int fork(mi355_bctx* c, hipStream_t s, hipStream_t* w) {
  if (!c->overlap) {
    *w = s;
    return 0;
  }
  hipEvent_t e = c->ev[c->ev_next++ % c->ev.size()];
  MI355_HIP(hipEventRecord(e, s));
  MI355_HIP(hipStreamWaitEvent(c->wstream, e, 0));
  c->w_dirty = true;
  *w = c->wstream;
  return 0;
}
int join(mi355_bctx* c, hipStream_t s) {
  if (!c->overlap || !c->w_dirty) return 0;
  hipEvent_t e = c->ev[c->ev_next++ % c->ev.size()];
  MI355_HIP(hipEventRecord(e, c->wstream));
  MI355_HIP(hipStreamWaitEvent(s, e, 0));
  c->w_dirty = false;
  return 0;
}

// ---- BN parameter views: the flat arrays directly, or the zero / one padded staging rows of the 32-channel stem BNs ---------------
struct BNP {
  float *g, *b, *rm, *rv, *dg, *db;
};
int bn_params(mi355_bctx* c, VBN& b, BNP& q, bool load, hipStream_t s) {
  if (b.Cp == b.C) {
    q = BNP{c->params + b.g_off, c->params + b.b_off, c->buffers + b.rm_off, c->buffers + b.rv_off, c->grads + b.g_off, c->grads + b.b_off};
    return 0;
  }
  float* st = b.stage;
  q = BNP{st, st + b.Cp, st + 2 * b.Cp, st + 3 * b.Cp, st + 4 * b.Cp, st + 5 * b.Cp};
  if (load) {  // (the padded tails were set at creation: gamma / beta / mean 0, variance 1 => the padded channels stay 0)
    const size_t n = (size_t)b.C * 4;
    MI355_HIP(hipMemcpyAsync(q.g, c->params + b.g_off, n, hipMemcpyDeviceToDevice, s));
    MI355_HIP(hipMemcpyAsync(q.b, c->params + b.b_off, n, hipMemcpyDeviceToDevice, s));
    MI355_HIP(hipMemcpyAsync(q.rm, c->buffers + b.rm_off, n, hipMemcpyDeviceToDevice, s));
    MI355_HIP(hipMemcpyAsync(q.rv, c->buffers + b.rv_off, n, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

int prep_weight(mi355_bctx* c, VConv& v, hipStream_t s) {
  const float* w = c->params + v.w_off;
  if (c->wstd) {
    MI355_TRY(mi355_weight_std_fwd(w, v.w_hat, v.ws_mean, v.ws_invstd, v.Cout, v.K * v.K * v.Cin, WS_EPS, s));
    w = v.w_hat;
  }
  return launch_weight_pad_cast(c->dtype, w, v.wp, v.Cout, v.K * v.K, v.Cin, v.Coutp, v.Cinp, s);
}

// y = conv(in), out = act(bn(y)): the conv epilogue sums y for the batch statistics wherever its launch shape can
// lazy_ok (identity-activation BNs whose only consumer is the fused ECA / residual pass): leave the normalisation to that pass — b.scale /
// b.shift are written, b.out is not (b.lazy says which happened: a conv launch that could not carry the statistics takes the plain route)
int conv_bn(mi355_bctx* c, VConv& v, VBN& b, const void* in, bool training, float momentum, hipStream_t s, bool lazy_ok = false) {
  const int M = c->N * v.Hout * v.Wout;
  BNP q;
  MI355_TRY(bn_params(c, b, q, true, s));
  b.lazy = false;
  if (!training) {
    MI355_TRY(mi355_conv2d_fwd(c->dtype, in, v.wp, v.y, c->N, v.Hin, v.Win, v.Cinp, v.Coutp, v.K, v.K, v.stride, v.pad, s));
    if (lazy_ok) {
      b.lazy = true;
      return launch_bn_eval_coeffs(q.g, q.b, q.rm, q.rv, b.scale, b.shift, b.Cp, BN_EPS, s);
    }
    return mi355_bn_fwd_eval(c->dtype, v.y, nullptr, b.out, q.g, q.b, q.rm, q.rv, M, b.Cp, BN_EPS, b.act, c->bn_ws, c->bn_ws_bytes, s);
  }
  int nblk = 0;
  MI355_TRY(mi355_conv2d_fwd_stats(c->dtype, in, v.wp, v.y, c->partial, c->partial_bytes, &nblk, c->N, v.Hin, v.Win, v.Cinp, v.Coutp, v.K, v.K,
                                   v.stride, v.pad, s));
  b.has_bits = false;
  if (nblk > 0) {
    // finalize from the conv epilogue's partial rows, then normalise + activate; the activation's sign goes out as one bit per element
    // (backward then reads 1/16 of a tensor instead of `out` itself, in both of its passes)
    MI355_TRY(launch_bn_finalize(c->partial, nullptr, nblk, M, b.Cp, q.g, q.b, q.rm, q.rv, b.mean, b.invstd, b.scale, b.shift, BN_EPS, momentum, s));
    if (lazy_ok) {
      b.lazy = true;
    } else {
      uint8_t* bits = (c->use_bits && b.act != ACT_NONE) ? b.bits : nullptr;
      MI355_TRY(launch_bn_apply(c->dtype, v.y, b.scale, b.shift, nullptr, nullptr, nullptr, nullptr, b.out, M, b.Cp, b.act, s, bits));
      b.has_bits = bits != nullptr;
    }
  } else {
    MI355_TRY(mi355_bn_fwd_train(c->dtype, v.y, nullptr, b.out, q.g, q.b, q.rm, q.rv, b.mean, b.invstd, M, b.Cp, BN_EPS, momentum, b.act, c->bn_ws,
                                 c->bn_ws_bytes, s));
  }
  if (b.Cp != b.C) {  // running statistics back into the flat buffer array
    MI355_HIP(hipMemcpyAsync(c->buffers + b.rm_off, q.rm, (size_t)b.C * 4, hipMemcpyDeviceToDevice, s));
    MI355_HIP(hipMemcpyAsync(c->buffers + b.rv_off, q.rv, (size_t)b.C * 4, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// v.dy = gradient wrt the conv output, from the gradient wrt the activation b.dout (dout may be another buffer)
// eg (optional): `dout` is the gradient wrt an ECA module's OUTPUT side (see EcaGrad): both passes form the module's input gradient on the fly
// have_rows > 0: the partial rows of the sums are already in c->bn_ws (left by the fused ECA backward or by the data gradient that produced dout): no reduction pass
// rows_at: ... kept elsewhere (the downsample BatchNorm's row waits there while other layers use c->bn_ws)
int bn_back(mi355_bctx* c, VConv& v, VBN& b, const void* dout, float beta, hipStream_t s, const EcaGrad* eg = nullptr, int have_rows = 0,
            const float* rows_at = nullptr) {
  const int M = c->N * v.Hout * v.Wout;
  BNP q;
  MI355_TRY(bn_params(c, b, q, false, s));  // (the staged gamma of this step's forward is still there)
  const bool staged = b.Cp != b.C;
  {
    float* partial = (float*)c->bn_ws;
    float* coef = partial + (size_t)std::max(bn_max_blocks(), have_rows) * 2 * b.Cp;   // (a data gradient's epilogue may leave more rows than a reduction pass)
    const uint8_t* bits = b.has_bits ? b.bits : nullptr;
    const void* mask = (b.act != ACT_NONE && !bits) ? b.out : nullptr;
    const float slope = b.act == ACT_LEAKY ? 0.01f : 0.f;
    int nblk = have_rows;
    if (nblk == 0) MI355_TRY(launch_bn_bwd_reduce(c->dtype, dout, mask, v.y, b.mean, b.invstd, nullptr, partial, &nblk, M, b.Cp, s, bits, slope, eg));
    MI355_TRY(launch_bn_bwd_finalize(have_rows && rows_at ? rows_at : partial, nblk, M, b.Cp, q.g, b.invstd, q.dg, q.db, staged ? 0.f : beta, coef, s));
    MI355_TRY(launch_bn_bwd_apply(c->dtype, dout, mask, v.y, b.mean, b.invstd, coef, v.dy, M, b.Cp, s, bits, slope, QuantOut(), eg));
  }
  if (staged) {
    MI355_TRY(launch_axpby(q.dg, c->grads + b.g_off, beta, (size_t)b.C, s));
    MI355_TRY(launch_axpby(q.db, c->grads + b.b_off, beta, (size_t)b.C, s));
  }
  return 0;
}

// weight gradient of v from v.dy and its input, on the side stream: wgrad -> [un-pad] -> [standardisation backward] -> flat gradients
int conv_wgrad(mi355_bctx* c, VConv& v, const void* in, float beta, hipStream_t s) {
  hipStream_t w;
  MI355_TRY(fork(c, s, &w));
  float* g = c->grads + v.w_off;
  const bool direct = !v.padded() && !c->wstd;
  float* dwp = direct ? g : c->dw_tmp;
  MI355_TRY(mi355_conv2d_wgrad(c->dtype, v.dy, in, dwp, direct ? beta : 0.f, c->N, v.Hin, v.Win, v.Cinp, v.Coutp, v.K, v.K, v.stride, v.pad, c->wg_ws,
                               c->wg_ws_bytes, w));
  if (direct) return 0;
  const float* dwh = dwp;
  if (v.padded()) {
    float* dst = c->wstd ? c->dw_tmp + v.wpelems() : g;  // (behind the padded gradient in the same scratch)
    MI355_TRY(launch_weight_unpad(dwp, dst, c->wstd ? 0.f : beta, v.Cout, v.K * v.K, v.Cin, v.Cinp, w));
    dwh = dst;
  }
  if (c->wstd) MI355_TRY(mi355_weight_std_bwd(dwh, v.w_hat, v.ws_invstd, g, beta, v.Cout, v.K * v.K * v.Cin, w));
  return 0;
}

// dx = conv_transpose(v.dy) (+ addend), from the transposed weights the forward's weight preparation left in v.wtr (mi355_conv2d_dgrad
// would transpose them again in front of every launch: 54 small kernels on backward's critical path)
// pv / pb / rows: dx is the activation gradient of the BatchNorm pb behind the convolution pv — where a generated kernel has the epilogue
// (asm/dconv_gen.py Cfg.stats == 3: the sums under the leaky-ReLU bit mask), the BN-backward sums of pb ride in this launch: *rows partial rows in
// c->bn_ws, and bn_back() runs without its reduction pass (2 x the tensor + its mask not read again); *rows == 0: not fused
int conv_dgrad(mi355_bctx* c, VConv& v, void* dx, const void* addend, hipStream_t s, VConv* pv = nullptr, VBN* pb = nullptr, int* rows = nullptr) {
  IgemmArgs a;
  const int nclass = build_dgrad_args(a, c->N, v.Hin, v.Win, v.Cinp, v.Coutp, v.K, v.K, v.stride, v.pad);
  if (nclass < 0) return nclass;
  a.in = v.dy; a.wt = v.wtr; a.out = dx; a.addend = addend;
  if (rows) *rows = 0;
  if (pv && pb && rows && c->fuse_bn_bwd && pb->has_bits && pb->act == ACT_LEAKY && pb->Cp == v.Cinp) {
    IgemmArgs t = a;
    t.bn_y = pv->y; t.bn_bits = pb->bits; t.bn_mean = pb->mean; t.bn_invstd = pb->invstd; t.bn_slope = 0.01f;
    t.stat_partial = (float*)c->bn_ws;
    t.stat_rows_cap = (int)std::min<size_t>((c->bn_ws_bytes / 4 - (size_t)3 * pb->Cp) / ((size_t)2 * pb->Cp), 1u << 20);
    if (igemm_leaky_sums_legal(c->dtype, t, nclass)) return launch_igemm(c->dtype, t, nclass, s, rows);
  }
  return launch_igemm(c->dtype, a, nclass, s);
}

size_t conv_ws_bytes(const mi355_bctx* c, const VConv& v) {
  return mi355_conv2d_workspace_bytes(c->dtype, c->N, v.Hin, v.Win, v.Cinp, v.Coutp, v.K, v.K, v.stride, v.pad);
}

template <typename F>
void for_each_conv(mi355_bctx* c, F f) {
  f(c->s0); f(c->s1); f(c->s2);
  for (VBlock& b : c->blocks) {
    f(b.c1); f(b.c2); f(b.c3);
    if (b.has_ds) f(b.ds);
  }
}

}  // namespace

// consecutive backward segments form buckets of >= cap_elems gradient elements; a bucket is the contiguous span of its segments.  The
// LAST bucket (nothing left to overlap its all-reduce with) is cut once more: its trailing segments up to cap_elems / 8 form a bucket
// of their own (the rule of resnet_exec.cpp plan_buckets and parallel.plan_buckets, stated for any segment order).
static std::vector<mi355_bctx::Bucket> plan_bbuckets(const mi355_bctx* c, size_t cap_elems) {
  std::vector<mi355_bctx::Bucket> out;
  std::vector<int> firsts;
  const int nseg = (int)c->segs.size();
  auto span = [&](int f, int l) {
    size_t lo = c->segs[f].first, hi = c->segs[f].second;
    for (int i = f; i <= l; ++i) { lo = std::min(lo, c->segs[i].first); hi = std::max(hi, c->segs[i].second); }
    return mi355_bctx::Bucket{lo, hi, l};
  };
  int first = -1;
  size_t size = 0;
  for (int i = 0; i < nseg; ++i) {
    if (first < 0) { first = i; size = 0; }
    size += c->segs[i].second - c->segs[i].first;
    if (size >= cap_elems || i == nseg - 1) {
      out.push_back(span(first, i));
      firsts.push_back(first);
      first = -1;
    }
  }
  const size_t tail_cap = cap_elems / 8;
  if (!out.empty() && out.back().end - out.back().begin > tail_cap) {
    const int f = firsts.back(), l = out.back().last_seg;
    int cut = l + 1;
    size_t tail = 0;
    for (int i = l; i > f; --i) {
      const size_t n = c->segs[i].second - c->segs[i].first;
      if (tail + n > tail_cap) break;
      tail += n;
      cut = i;
    }
    if (cut > f && cut <= l) {
      out.back() = span(f, cut - 1);
      out.push_back(span(cut, l));
    }
  }
  return out;
}

// segment `seg` of this backward call is enqueued on both streams: reduce the buckets it completes
static int after_segment(mi355_bctx* c, int seg, hipStream_t s) {
  if (!c->comm || !c->grad_sync) return 0;
  for (const auto& bk : c->buckets)
    if (bk.last_seg == seg) {
      MI355_TRY(comm_allreduce_bucket(c->comm, c->grads, bk.begin, bk.end, s, c->overlap && c->w_dirty ? c->wstream : nullptr));
      c->comm_dirty = true;
    }
  return 0;
}

extern "C" {

int mi355_bresnet50_create(mi355_bctx** out, int device, int dtype, int N, int H, int W, int num_classes, int weight_std) {
  MI355_ARG(out, "bresnet50_create: null out");
  MI355_ARG(dtype == MI355_F32 || dtype == MI355_BF16, "bresnet50_create: bad dtype %d", dtype);
  MI355_ARG(N >= 1 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, "bresnet50_create: N=%d H=%d W=%d (H, W multiples of 32)", N, H, W);
  MI355_ARG(num_classes >= 1 && num_classes <= 65536, "bresnet50_create: num_classes=%d", num_classes);
  if (device >= 0) MI355_HIP(hipSetDevice(device));
  mi355_bctx* c = new mi355_bctx();
  c->device = device; c->dtype = dtype; c->N = N; c->H = H; c->W = W; c->num_classes = num_classes;
  c->es = dtype_size(dtype);
  c->wstd = weight_std != 0;
  c->fc_pad = (int)align_up((size_t)num_classes, 128);
  // ---- graph + tensor table (pytorch_tools names, registration order of bresnet.py) -------------------------------------------
  init_conv(c, c->s0, "conv1.0", 3, 32, 3, 2, H, W);
  {
    c->stem_im2col = env_switch("MI355_BRESNET_STEM_IM2COL", 1, 1, &c->bad_switch) != 0;
    if (c->stem_im2col) {
      // the same parameter tensor [32][3][3][3] read as [32][27], the same output grid: a pointwise convolution over the patch tensor the input
      // conversion writes (launch_nchw_im2col3s2)
      VConv& v = c->s0;
      v.Cin = 27; v.K = 1; v.stride = 1; v.pad = 0; v.Hin = v.Hout; v.Win = v.Wout; v.Cinp = 64;
    }
  }
  init_bn(c, c->sb0, "conv1.1", 32, ACT_LEAKY);
  init_conv(c, c->s1, "conv1.2", 32, 32, 3, 1, H / 2, W / 2);
  init_bn(c, c->sb1, "conv1.3", 32, ACT_LEAKY);
  init_conv(c, c->s2, "conv1.4", 32, 64, 3, 1, H / 2, W / 2);
  init_bn(c, c->sb2, "bn1", 64, ACT_LEAKY);
  int h = H / 4, w = W / 4, cin = 64;
  const int nb[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
  for (int li = 0; li < 4; ++li)
    for (int i = 0; i < nb[li]; ++i) {
      VBlock b;
      const std::string pre = "layer" + std::to_string(li + 1) + "." + std::to_string(i);
      b.cin = cin; b.planes = planes[li];
      b.stride = (i == 0 && li > 0) ? 2 : 1;
      b.has_ds = i == 0;
      b.H = h; b.W = w; b.Ho = h / b.stride; b.Wo = w / b.stride;
      init_conv(c, b.c1, pre + ".conv1", cin, b.planes, 1, 1, h, w);
      init_bn(c, b.b1, pre + ".bn1", b.planes, ACT_LEAKY);
      init_conv(c, b.c2, pre + ".conv2", b.planes, b.planes, 3, 1, h, w);  // stride 1: the block strides through its BlurPool
      init_bn(c, b.b2, pre + ".bn2", b.planes, ACT_LEAKY);
      init_conv(c, b.c3, pre + ".conv3", b.planes, 4 * b.planes, 1, 1, b.Ho, b.Wo);
      init_bn(c, b.b3, pre + ".bn3", 4 * b.planes, ACT_NONE);
      b.eca_off = reg(c, pre + ".se_module.conv.weight", 0, {1, 1, 3});
      if (b.has_ds) {
        init_conv(c, b.ds, pre + ".downsample.0", cin, 4 * b.planes, 1, 1, b.Ho, b.Wo);
        init_bn(c, b.bd, pre + ".downsample.1", 4 * b.planes, ACT_NONE);
      }
      c->blocks.push_back(b);
      h = b.Ho; w = b.Wo; cin = 4 * b.planes;
    }
  // FC rows padded to a multiple of 128 inside the flat array (the tail rows stay zero), as in the ResNet-50 executor
  c->fc_w_off = reg(c, "fc.weight", 0, {num_classes, 2048});
  c->param_elems += (size_t)(c->fc_pad - num_classes) * 2048;
  c->fc_b_off = reg(c, "fc.bias", 0, {num_classes});
  c->param_elems = align_up(c->param_elems, 64);
  c->buffer_elems = align_up(c->buffer_elems, 64);
  c->fwd_flops += 2.0 * N * 2048.0 * num_classes;
  // backward segments over the flat gradient array
  {
    const int nb = (int)c->blocks.size();
    c->segs.push_back({c->fc_w_off, c->param_elems});
    for (int i = nb - 1; i >= 0; --i) c->segs.push_back({c->blocks[i].c1.w_off, i + 1 < nb ? c->blocks[i + 1].c1.w_off : c->fc_w_off});
    c->segs.push_back({0, c->blocks[0].c1.w_off});
  }
  *out = c;
  if (device < 0) return 0;  // layout-only context (no GPU): tensor table and sizes

  // ---- workspace ---------------------------------------------------------------------------------------------------------------
  Arena ar;
  const size_t es = c->es;
  auto act_bytes = [&](int hh, int ww, int ch) { return (size_t)N * hh * ww * ch * es; };
  ar.add(&c->h0, c->stem_im2col ? act_bytes(H / 2, W / 2, 64) : act_bytes(H, W, 64));
  plan_conv(c, ar, c->s0); plan_bn(c, ar, c->sb0, H / 2, W / 2);
  plan_conv(c, ar, c->s1); plan_bn(c, ar, c->sb1, H / 2, W / 2);
  plan_conv(c, ar, c->s2); plan_bn(c, ar, c->sb2, H / 2, W / 2);
  ar.add(&c->m, act_bytes(H / 2, W / 2, 64));
  ar.add(&c->dm, act_bytes(H / 2, W / 2, 64));
  ar.add(&c->pool_idx, (size_t)N * (H / 2) * (W / 2) * 64);
  ar.add(&c->p, act_bytes(H / 4, W / 4, 64));
  for (VBlock& b : c->blocks) {
    plan_conv(c, ar, b.c1); plan_bn(c, ar, b.b1, b.H, b.W);
    plan_conv(c, ar, b.c2); plan_bn(c, ar, b.b2, b.H, b.W);
    if (b.stride == 2) {
      ar.add(&b.a2b, act_bytes(b.Ho, b.Wo, b.planes));
      ar.add(&b.da2b, act_bytes(b.Ho, b.Wo, b.planes));
      ar.add(&b.scin, act_bytes(b.Ho, b.Wo, b.cin));
      ar.add(&b.dscin, act_bytes(b.Ho, b.Wo, b.cin));
    }
    plan_conv(c, ar, b.c3); plan_bn(c, ar, b.b3, b.Ho, b.Wo);
    const size_t o = act_bytes(b.Ho, b.Wo, 4 * b.planes);
    ar.add(&b.e, o);
    ar.add(&b.de, o);
    ar.add(&b.pooled, (size_t)N * 4 * b.planes * 4);
    ar.add(&b.gate, (size_t)N * 4 * b.planes * 4);
    ar.add(&b.keep, (size_t)N * 4);
    if (b.has_ds) {
      plan_conv(c, ar, b.ds); plan_bn(c, ar, b.bd, b.Ho, b.Wo, false);
    }
    ar.add(&b.out, o);
    ar.add(&b.out_bits, o / 16);
    ar.add(&b.dsc, o);
    const size_t ib = act_bytes(b.H, b.W, b.cin);
    ar.add(&b.dxb, ib);
    ar.add(&b.dxs, ib);
    ar.add(&b.dx, ib);
  }
  ar.add(&c->dlast, act_bytes(H / 32, W / 32, 2048));
  ar.add(&c->pooled, (size_t)N * 2048 * 4);
  ar.add(&c->pooled_d, (size_t)N * 2048 * 4);
  ar.add(&c->do_mask, (size_t)N * 2048 * 4);
  ar.add(&c->dpooled, (size_t)N * 2048 * 4);
  ar.add(&c->fc_tmp, (size_t)N * c->fc_pad * 4);
  ar.add(&c->dlogits_pad, (size_t)N * c->fc_pad * 4);
  ar.add(&c->fc_wtr, (size_t)c->fc_pad * 2048 * 4);
  c->partial_bytes = (size_t)768 * 2 * 2048 * 4;
  ar.add(&c->partial, c->partial_bytes);
  c->bn_ws_bytes = mi355_bn_workspace_bytes(2048);
  ar.add(&c->bn_ws, c->bn_ws_bytes);
  size_t ws_max = 0, dw_max = 0;
  for_each_conv(c, [&](VConv& v) {
    ws_max = std::max(ws_max, conv_ws_bytes(c, v));
    dw_max = std::max(dw_max, v.wpelems() + v.welems());
  });
  const size_t fc_wg = (size_t)plan_wgrad_splits(MI355_F32, N, c->fc_pad, 1, 2048) * c->fc_pad * 2048 * 4;
  c->wg_ws_bytes = std::max(ws_max, fc_wg);
  ar.add(&c->wg_ws, c->wg_ws_bytes);
  ar.add(&c->dw_tmp, dw_max * 4);
  ar.add(&c->eca_ws, ((size_t)6 * N * 2048 + 1152 + 2 * 2048) * 4);  // [s][dpool][dw parts][4 per-image sums for bn3's / the downsample BN's backward][the latter's row]
  ar.add((void**)&c->prep_table, (size_t)64 * sizeof(BPrepDesc));
  c->arena_bytes = ar.size;
  if (hipMalloc((void**)&c->arena, c->arena_bytes) != hipSuccess) {
    set_error("bresnet50_create: hipMalloc(%zu bytes) failed: %s", c->arena_bytes, hipGetErrorString(hipGetLastError()));
    delete c;
    *out = nullptr;
    return MI355_E_HIP;
  }
  for (auto& sl : ar.slots) *sl.first = c->arena + sl.second;
  // padded BN staging rows: gamma / beta / running_mean 0, running_var 1 (the per-op graph padded them the same way)
  for (VBN* b : {&c->sb0, &c->sb1})
    if (b->stage) {
      std::vector<float> st((size_t)6 * b->Cp, 0.f);
      for (int i = 0; i < b->Cp; ++i) st[(size_t)3 * b->Cp + i] = 1.f;
      if (hipMemcpy(b->stage, st.data(), st.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("bresnet50_create: staging upload failed");
        mi355_bresnet50_destroy(c);
        *out = nullptr;
        return MI355_E_HIP;
      }
    }
  // executor switches: closed domains, read here once (the bit-identity tests against the per-op graph switch the fused forms off)
  c->overlap = env_switch("MI355_WGRAD_STREAM", 1, 1, &c->bad_switch) != 0;
  c->fused_add = env_switch("MI355_BRESNET_FUSED_ADD", 1, 1, &c->bad_switch) != 0;
  c->fuse_bn_bwd = env_switch("MI355_BRESNET_FUSE_BN_BWD", 1, 1, &c->bad_switch) != 0;
  c->batch_prep = env_switch("MI355_BRESNET_BATCH_PREP", 1, 1, &c->bad_switch) != 0;
  c->lazy_bn = env_switch("MI355_BRESNET_LAZY_BN", 1, 1, &c->bad_switch) != 0;
  c->fused_eca = env_switch("MI355_BRESNET_FUSED_ECA", 1, 1, &c->bad_switch) != 0;
  c->eca_sums = env_switch("MI355_BRESNET_ECA_SUMS", 1, 1, &c->bad_switch) != 0;
  if (c->bad_switch) {
    set_error("bresnet50_create: %s=%s is outside the switch's domain", c->bad_switch, getenv(c->bad_switch));
    mi355_bresnet50_destroy(c);
    *out = nullptr;
    return MI355_E_ARG;
  }
  bool ok = true;
  if (c->overlap) {
    ok = hipStreamCreateWithFlags(&c->wstream, hipStreamNonBlocking) == hipSuccess;
    c->ev.resize(64);
    for (auto& e : c->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {
    set_error("bresnet50_create: stream / event creation failed");
    mi355_bresnet50_destroy(c);
    *out = nullptr;
    return MI355_E_HIP;
  }
  return 0;
}

int mi355_bresnet50_destroy(mi355_bctx* c) {
  if (!c) return 0;
  if (c->wstream) {
    (void)hipStreamSynchronize(c->wstream);
    (void)hipStreamDestroy(c->wstream);
  }
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->arena) (void)hipFree(c->arena);
  delete c;
  return 0;
}

int mi355_bresnet50_num_tensors(const mi355_bctx* c) { return c ? (int)c->tensors.size() : 0; }

int mi355_bresnet50_tensor_info(const mi355_bctx* c, int idx, char* name, int name_cap, int* kind, size_t* offset, int* ndim, int* shape) {
  MI355_ARG(c && idx >= 0 && idx < (int)c->tensors.size() && name && name_cap > 0, "bresnet50_tensor_info: bad index %d", idx);
  const TInfo& t = c->tensors[idx];
  snprintf(name, (size_t)name_cap, "%s", t.name.c_str());
  if (kind) *kind = t.kind;
  if (offset) *offset = t.off;
  if (ndim) *ndim = t.ndim;
  if (shape)
    for (int i = 0; i < t.ndim; ++i) shape[i] = t.shape[i];
  return 0;
}

size_t mi355_bresnet50_flat_param_elems(const mi355_bctx* c) { return c ? c->param_elems : 0; }
size_t mi355_bresnet50_flat_buffer_elems(const mi355_bctx* c) { return c ? c->buffer_elems : 0; }
size_t mi355_bresnet50_workspace_bytes(const mi355_bctx* c) { return c ? c->arena_bytes : 0; }

int mi355_bresnet50_bind(mi355_bctx* c, float* params, float* grads, float* buffers) {
  MI355_ARG(c && params && grads && buffers, "bresnet50_bind: null pointer");
  c->params = params; c->grads = grads; c->buffers = buffers;
  return 0;
}

int mi355_bresnet50_set_drop(mi355_bctx* c, float drop_rate, float drop_connect_rate, unsigned long long seed) {
  MI355_ARG(c && drop_rate >= 0.f && drop_rate < 1.f && drop_connect_rate >= 0.f && drop_connect_rate < 1.f, "bresnet50_set_drop: rates in [0, 1)");
  c->drop_rate = drop_rate; c->drop_connect = drop_connect_rate; c->seed = seed;
  return 0;
}

int mi355_bresnet50_flops(const mi355_bctx* c, double* fwd, double* train) {
  MI355_ARG(c, "bresnet50_flops: null ctx");
  // training = forward + dgrad + wgrad of every conv (the first conv has no dgrad) + the FC's three GEMMs
  const double first = 2.0 * c->N * c->s0.Hout * c->s0.Wout * (double)c->s0.Cout * c->s0.Cin * 9;
  if (fwd) *fwd = c->fwd_flops;
  if (train) *train = 3.0 * c->fwd_flops - first;
  return 0;
}

int mi355_bresnet50_forward(mi355_bctx* c, const float* x_nchw, float* logits, int training, float bn_momentum, unsigned long long step,
                            const float* const* keep_override, const float* dropout_override, void* stream) {
  MI355_ARG(c && c->arena && c->params, "bresnet50_forward: context not bound / layout-only");
  MI355_ARG(x_nchw && logits, "bresnet50_forward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int N = c->N, dt = c->dtype;
  const bool tr = training != 0;
  c->have_fwd = false;  // (until this pass has completed: a failed forward must not be differentiated)
  MI355_TRY(join(c, s));  // (a previous backward's side-stream work reads the tensors this pass overwrites)
  // ---- weight preparation: [standardise] -> pad -> cast; FC weights transposed for the input gradient ------------------------------
  // The first conv's weights on the caller's stream; everything else on the side stream, beside the input conversion and the first conv, which only
  // have to wait for their own: two launches over a table of every layer (variant.hip launch_bres_weight_prep) — as 106 + 54 small launches their
  // dispatch chain outlasted the first conv and held the second one up.
  MI355_TRY(prep_weight(c, c->s0, s));
  hipStream_t ps;
  MI355_TRY(fork(c, s, &ps));
  int rc = 0;
  if (c->batch_prep) {
    if (c->prep_params != c->params) {   // (once per bound parameter array: the table holds pointers into it)
      std::vector<BPrepDesc> tab;
      int rows = 0, tiles = 0;
      for_each_conv(c, [&](VConv& v) {
        if (&v == &c->s0) return;
        BPrepDesc d{};
        d.w = c->params + v.w_off; d.w_hat = c->wstd ? v.w_hat : nullptr; d.mean = v.ws_mean; d.invstd = v.ws_invstd; d.wp = v.wp; d.wtr = v.wtr;
        d.Cout = v.Cout; d.taps = v.K * v.K; d.Cin = v.Cin; d.Coutp = v.Coutp; d.Cinp = v.Cinp; d.row_begin = rows; d.tile_begin = tiles;
        rows += v.Coutp;
        tiles += (v.Cinp / 32) * (v.Coutp / 32) * v.K * v.K;
        if (v.Cinp % 32 || v.Coutp % 32) rc = MI355_E_ARG;
        tab.push_back(d);
      });
      MI355_ARG(rc == 0 && tab.size() <= 64, "bresnet50_forward: weight table (%zu layers)", tab.size());
      MI355_HIP(hipMemcpy(c->prep_table, tab.data(), tab.size() * sizeof(BPrepDesc), hipMemcpyHostToDevice));
      c->prep_params = c->params; c->prep_n = (int)tab.size(); c->prep_rows = rows; c->prep_tiles = tiles;
    }
    MI355_TRY(launch_bres_weight_prep(dt, c->prep_table, c->prep_n, c->prep_rows, c->prep_tiles, WS_EPS, tr, ps));
  } else {
    for_each_conv(c, [&](VConv& v) {
      if (rc == 0 && &v != &c->s0) rc = prep_weight(c, v, ps);
      if (rc == 0 && tr && &v != &c->s0) rc = launch_transpose_any(dt, v.wp, v.wtr, v.Coutp, v.K * v.K, v.Cinp, ps);
    });
  }
  if (rc) return rc;
  if (tr) MI355_TRY(launch_weight_prep(MI355_F32, c->params + c->fc_w_off, nullptr, c->fc_wtr, c->fc_pad, 1, 2048, ps));
  // the drop-connect scales of every block and the dropout mask: one launch here instead of one per block on the caller's stream (counter-based: the same values)
  const bool batch_keep = c->batch_prep && !keep_override && tr;
  if (batch_keep) {
    KeepBatch kb;
    const int nb = (int)c->blocks.size();
    for (int i = 1; i < nb && c->drop_connect > 0.f; ++i) {
      const int e = kb.count++;
      kb.keep[e] = c->blocks[i].keep; kb.n[e] = (size_t)N; kb.p[e] = c->drop_connect * (float)i / (float)nb; kb.counter[e] = step * 64 + (unsigned long long)i;
    }
    if (!dropout_override && c->drop_rate > 0.f) {
      const int e = kb.count++;
      kb.keep[e] = c->do_mask; kb.n[e] = (size_t)N * 2048; kb.p[e] = c->drop_rate; kb.counter[e] = step * 64 + 63;
    }
    MI355_TRY(launch_keep_scale_batch(kb, c->seed, ps));
  }
  // ---- stem --------------------------------------------------------------------------------------------------------------------
  if (c->stem_im2col) MI355_TRY(launch_nchw_im2col3s2(dt, x_nchw, c->h0, N, c->H, c->W, s));
  else MI355_TRY(launch_nchw_pad64(dt, x_nchw, c->h0, N, c->H * c->W, s));
  MI355_TRY(conv_bn(c, c->s0, c->sb0, c->h0, tr, bn_momentum, s));
  MI355_TRY(join(c, s));  // the prepared weights of every later layer
  MI355_TRY(conv_bn(c, c->s1, c->sb1, c->sb0.out, tr, bn_momentum, s));
  MI355_TRY(conv_bn(c, c->s2, c->sb2, c->sb1.out, tr, bn_momentum, s));
  MI355_TRY(mi355_maxpool3s1_fwd(dt, c->sb2.out, c->m, c->pool_idx, N, c->H / 2, c->W / 2, 64, s));
  MI355_TRY(mi355_blurpool_fwd(dt, c->m, c->p, N, c->H / 2, c->W / 2, 64, s));
  // ---- bottlenecks -------------------------------------------------------------------------------------------------------------
  const void* x = c->p;
  const int nblocks = (int)c->blocks.size();
  for (int i = 0; i < nblocks; ++i) {
    VBlock& b = c->blocks[i];
    MI355_TRY(conv_bn(c, b.c1, b.b1, x, tr, bn_momentum, s));
    MI355_TRY(conv_bn(c, b.c2, b.b2, b.b1.out, tr, bn_momentum, s));
    const void* a2 = b.b2.out;
    if (b.stride == 2) {
      MI355_TRY(mi355_blurpool_fwd(dt, a2, b.a2b, N, b.H, b.W, b.planes, s));
      a2 = b.a2b;
    }
    const bool lazy_ok = c->fused_eca && c->lazy_bn;  // bn3 / the downsample BN are applied inside the fused ECA + residual pass
    MI355_TRY(conv_bn(c, b.c3, b.b3, a2, tr, bn_momentum, s, lazy_ok));
    const int C4 = 4 * b.planes;
    if (!c->fused_eca) MI355_TRY(mi355_eca_fwd(dt, b.b3.out, c->params + b.eca_off, 3, b.e, b.pooled, b.gate, N, b.Ho * b.Wo, C4, s));
    const void* sc = x;
    if (b.has_ds) {
      const void* scin = x;
      if (b.stride == 2) {
        MI355_TRY(mi355_avgpool2_fwd(dt, x, b.scin, N, b.H, b.W, b.cin, s));
        scin = b.scin;
      }
      MI355_TRY(conv_bn(c, b.ds, b.bd, scin, tr, bn_momentum, s, lazy_ok));
      sc = b.bd.lazy ? b.ds.y : b.bd.out;
    }
    // drop-connect: sample scales from the caller (tests) or from the counter-based generator, kept in b.keep for backward
    b.scaled = false;
    if (keep_override) {
      if (keep_override[i]) {
        MI355_HIP(hipMemcpyAsync(b.keep, keep_override[i], (size_t)N * 4, hipMemcpyDeviceToDevice, s));
        b.scaled = true;
      }
    } else if (tr && c->drop_connect > 0.f && i > 0) {
      if (!batch_keep) MI355_TRY(mi355_keep_scale(b.keep, (size_t)N, c->drop_connect * (float)i / (float)nblocks, c->seed, step * 64 + (unsigned long long)i, s));
      b.scaled = true;
    }
    b.has_out_bits = c->fused_eca && c->use_bits && tr;
    if (c->fused_eca)  // gate, drop-connect scale, shortcut add and activation in one pass: the gated tensor is never stored
      MI355_TRY(launch_eca_residual_fwd(dt, b.b3.lazy ? b.c3.y : b.b3.out, c->params + b.eca_off, 3, b.scaled ? b.keep : nullptr, sc, b.out, b.pooled, b.gate, N,
                                        b.Ho * b.Wo, C4, ACT_LEAKY, s, b.b3.lazy ? b.b3.scale : nullptr, b.b3.shift,
                                        (b.has_ds && b.bd.lazy) ? b.bd.scale : nullptr, b.bd.shift, b.has_out_bits ? b.out_bits : nullptr,
                                        c->batch_prep ? c->eca_ws : nullptr));
    else
      MI355_TRY(mi355_residual_act_fwd(dt, b.e, b.scaled ? b.keep : nullptr, sc, b.out, N, (size_t)b.Ho * b.Wo * C4, ACT_LEAKY, s));
    x = b.out;
  }
  // ---- head --------------------------------------------------------------------------------------------------------------------
  const VBlock& last = c->blocks.back();
  MI355_TRY(mi355_gap_fwd(dt, last.out, c->pooled, N, last.Ho * last.Wo, 2048, s));
  c->dropped = false;
  if (dropout_override) {
    MI355_HIP(hipMemcpyAsync(c->do_mask, dropout_override, (size_t)N * 2048 * 4, hipMemcpyDeviceToDevice, s));
    c->dropped = true;
  } else if (!keep_override && tr && c->drop_rate > 0.f) {
    if (!batch_keep) MI355_TRY(mi355_keep_scale(c->do_mask, (size_t)N * 2048, c->drop_rate, c->seed, step * 64 + 63, s));
    c->dropped = true;
  }
  const float* feat = c->pooled;
  if (c->dropped) {
    MI355_TRY(mi355_mul_f32(c->pooled, c->do_mask, c->pooled_d, (size_t)N * 2048, s));
    feat = c->pooled_d;
  }
  MI355_TRY(launch_fc(feat, c->params + c->fc_w_off, c->fc_tmp, N, c->fc_pad, 2048, s));
  MI355_TRY(launch_bias_slice(c->fc_tmp, c->fc_pad, c->params + c->fc_b_off, logits, N, c->num_classes, s));
  c->have_fwd = tr;
  return 0;
}

// Backward of the last training forward: every parameter gradient into the flat gradient array (accumulate != 0: added to what is
// there).  The weight-gradient convolutions run on the side stream; the call returns with both streams joined on `stream`.
int mi355_bresnet50_backward(mi355_bctx* c, const float* dlogits, int accumulate, void* stream) {
  MI355_ARG(c && c->arena && c->grads, "bresnet50_backward: context not bound / layout-only");
  MI355_ARG(c->have_fwd, "bresnet50_backward: no training forward to differentiate");
  MI355_ARG(dlogits, "bresnet50_backward: null dlogits");
  hipStream_t s = (hipStream_t)stream;
  const int N = c->N, dt = c->dtype, O = c->num_classes, P = c->fc_pad;
  const float beta = accumulate ? 1.f : 0.f;
  // ---- head: FC weight / bias gradient, input gradient, dropout, GAP ------------------------------------------------------------
  MI355_TRY(launch_pad_dlogits(dlogits, c->dlogits_pad, P, c->grads + c->fc_b_off, beta, N, O, s));
  {
    WgradArgs w;
    build_wgrad_args(w, N, 1, 1, 2048, P, 1, 1, 1, 0);
    w.dy = c->dlogits_pad; w.x = c->dropped ? c->pooled_d : c->pooled; w.partial = (float*)c->wg_ws;
    const int splits = plan_wgrad_splits(MI355_F32, N, P, 1, 2048);
    hipStream_t ws;
    MI355_TRY(fork(c, s, &ws));
    MI355_TRY(launch_wgrad(MI355_F32, w, splits, ws));
    MI355_TRY(launch_splitk_reduce((const float*)c->wg_ws, splits, (size_t)P * 2048, c->grads + c->fc_w_off, (size_t)O * 2048, beta, ws));
  }
  MI355_TRY(launch_fc(c->dlogits_pad, c->fc_wtr, c->dpooled, N, 2048, P, s));
  if (c->dropped) MI355_TRY(mi355_mul_f32(c->dpooled, c->do_mask, c->dpooled, (size_t)N * 2048, s));
  const VBlock& last = c->blocks.back();
  MI355_TRY(mi355_gap_bwd(dt, c->dpooled, c->dlast, N, last.Ho * last.Wo, 2048, s));
  MI355_TRY(after_segment(c, 0, s));
  // ---- bottlenecks, last to first ------------------------------------------------------------------------------------------------
  const void* g = c->dlast;  // gradient wrt the block output
  for (int i = (int)c->blocks.size() - 1; i >= 0; --i) {
    VBlock& b = c->blocks[i];
    const void* xin = i > 0 ? c->blocks[i - 1].out : c->p;
    const int C4 = 4 * b.planes;
    {  // test hooks: record / replace the gradient this segment starts from
      const size_t gbytes = (size_t)N * b.Ho * b.Wo * C4 * c->es;
      if (i < (int)c->hook_record.size() && c->hook_record[i]) MI355_HIP(hipMemcpyAsync(c->hook_record[i], g, gbytes, hipMemcpyDeviceToDevice, s));
      if (i < (int)c->hook_replay.size() && c->hook_replay[i]) MI355_HIP(hipMemcpyAsync(const_cast<void*>(g), c->hook_replay[i], gbytes, hipMemcpyDeviceToDevice, s));
    }
    bool lazy_dz = false, sums_row = false, ds_row = false;
    float* ds_row_at = c->eca_ws + (size_t)6 * N * 2048 + 1152;
    if (c->fused_eca) {
      // pass 2 of the ECA backward (dz3 = dsc * keep * gate + dpool) is left to bn3's backward, which forms it on the fly from dsc in
      // both of its passes: the gradient wrt bn3's output is never stored
      lazy_dz = c->lazy_dz3;
      sums_row = lazy_dz && b.b3.lazy && c->eca_sums && b.b3.Cp == C4;  // pass 1 leaves what bn3's backward needs from the tensors
      ds_row = sums_row && b.has_ds && b.bd.lazy && b.bd.Cp == C4;      // ... and the downsample BatchNorm's (its output gradient is dsc itself)
      MI355_TRY(launch_eca_residual_bwd(dt, g, b.out, b.b3.lazy ? b.c3.y : b.b3.out, b.scaled ? b.keep : nullptr, c->params + b.eca_off, 3, b.pooled, b.gate,
                                        b.dsc, lazy_dz ? nullptr : b.b3.dout, c->grads + b.eca_off, beta, c->eca_ws, N, b.Ho * b.Wo, C4, ACT_LEAKY, s,
                                        b.b3.lazy ? b.b3.scale : nullptr, b.b3.shift, sums_row ? (float*)c->bn_ws : nullptr, b.b3.mean, b.b3.invstd,
                                        b.has_out_bits ? b.out_bits : nullptr, ds_row ? b.ds.y : nullptr, ds_row ? ds_row_at : nullptr, b.bd.mean, b.bd.invstd));
    } else {
      MI355_TRY(mi355_residual_act_bwd(dt, g, b.out, b.scaled ? b.keep : nullptr, b.de, b.dsc, N, (size_t)b.Ho * b.Wo * C4, ACT_LEAKY, s));
      MI355_TRY(mi355_eca_bwd(dt, b.de, b.b3.out, c->params + b.eca_off, 3, b.pooled, b.gate, b.b3.dout, c->grads + b.eca_off, beta, c->eca_ws, N,
                              b.Ho * b.Wo, C4, s));
    }
    if (lazy_dz) {
      EcaGrad eg;
      eg.keep = b.scaled ? b.keep : nullptr; eg.gate = b.gate; eg.dpool = c->eca_ws + (size_t)N * C4; eg.hw = b.Ho * b.Wo;
      MI355_TRY(bn_back(c, b.c3, b.b3, b.dsc, beta, s, &eg, sums_row ? 1 : 0));
    } else {
      MI355_TRY(bn_back(c, b.c3, b.b3, b.b3.dout, beta, s));
    }
    const void* a2 = b.stride == 2 ? b.a2b : b.b2.out;
    int rows1 = 0, rows2 = 0;   // partial rows of bn1's / bn2's backward sums left by the data gradient in front of them (0: their own reduction pass)
    MI355_TRY(conv_wgrad(c, b.c3, a2, beta, s));
    if (b.stride == 2) {
      MI355_TRY(conv_dgrad(c, b.c3, b.da2b, nullptr, s));
      MI355_TRY(mi355_blurpool_bwd(dt, b.da2b, b.b2.dout, N, b.H, b.W, b.planes, s));
    } else {
      MI355_TRY(conv_dgrad(c, b.c3, b.b2.dout, nullptr, s, &b.c2, &b.b2, &rows2));   // + bn2's backward sums
    }
    MI355_TRY(bn_back(c, b.c2, b.b2, b.b2.dout, beta, s, nullptr, rows2));
    MI355_TRY(conv_wgrad(c, b.c2, b.b1.out, beta, s));
    MI355_TRY(conv_dgrad(c, b.c2, b.b1.dout, nullptr, s, &b.c1, &b.b1, &rows1));     // + bn1's backward sums
    MI355_TRY(bn_back(c, b.c1, b.b1, b.b1.dout, beta, s, nullptr, rows1));
    MI355_TRY(conv_wgrad(c, b.c1, xin, beta, s));
    // shortcut gradient first: it is the addend of conv1's input gradient
    const void* gs = b.dsc;
    if (b.has_ds) {
      MI355_TRY(bn_back(c, b.ds, b.bd, b.dsc, beta, s, nullptr, ds_row ? 1 : 0, ds_row_at));
      MI355_TRY(conv_wgrad(c, b.ds, b.stride == 2 ? b.scin : xin, beta, s));
      if (b.stride == 2) {
        MI355_TRY(conv_dgrad(c, b.ds, b.dscin, nullptr, s));
        MI355_TRY(mi355_avgpool2_bwd(dt, b.dscin, b.dxs, N, b.H, b.W, b.cin, s));
      } else {
        MI355_TRY(conv_dgrad(c, b.ds, b.dxs, nullptr, s));
      }
      gs = b.dxs;
    }
    if (c->fused_add) {
      MI355_TRY(conv_dgrad(c, b.c1, b.dx, gs, s));  // dx = dgrad + shortcut gradient in the conv epilogue (one rounding)
    } else {
      MI355_TRY(conv_dgrad(c, b.c1, b.dxb, nullptr, s));
      MI355_TRY(mi355_residual_act_fwd(dt, b.dxb, nullptr, gs, b.dx, N, (size_t)b.H * b.W * b.cin, ACT_NONE, s));  // (the autograd sum of the per-op graph)
    }
    g = b.dx;
    MI355_TRY(after_segment(c, (int)c->blocks.size() - i, s));
  }
  // ---- stem ------------------------------------------------------------------------------------------------------------------------
  MI355_TRY(mi355_blurpool_bwd(dt, g, c->dm, N, c->H / 2, c->W / 2, 64, s));
  MI355_TRY(mi355_maxpool3s1_bwd(dt, c->dm, c->pool_idx, c->sb2.dout, N, c->H / 2, c->W / 2, 64, s));
  MI355_TRY(bn_back(c, c->s2, c->sb2, c->sb2.dout, beta, s));
  MI355_TRY(conv_wgrad(c, c->s2, c->sb1.out, beta, s));
  int srows = 0;
  MI355_TRY(conv_dgrad(c, c->s2, c->sb1.dout, nullptr, s, &c->s1, &c->sb1, &srows));
  MI355_TRY(bn_back(c, c->s1, c->sb1, c->sb1.dout, beta, s, nullptr, srows));
  MI355_TRY(conv_wgrad(c, c->s1, c->sb0.out, beta, s));
  MI355_TRY(conv_dgrad(c, c->s1, c->sb0.dout, nullptr, s, &c->s0, &c->sb0, &srows));
  MI355_TRY(bn_back(c, c->s0, c->sb0, c->sb0.dout, beta, s, nullptr, srows));
  MI355_TRY(conv_wgrad(c, c->s0, c->h0, beta, s));
  MI355_TRY(after_segment(c, (int)c->blocks.size() + 1, s));
  MI355_TRY(join(c, s));
  if (c->comm && c->comm_dirty) {
    MI355_TRY(comm_join(c->comm, s));
    c->comm_dirty = false;
  }
  c->hook_record.clear();
  c->hook_replay.clear();
  return 0;
}

int mi355_bresnet50_grad_hooks(mi355_bctx* c, void* const* record, const void* const* replay, int nblocks) {
  MI355_ARG(c && c->arena && nblocks == (int)c->blocks.size(), "bresnet50_grad_hooks: %d entries for %zu blocks", nblocks, c ? c->blocks.size() : (size_t)0);
  c->hook_record.assign(nblocks, nullptr);
  c->hook_replay.assign(nblocks, nullptr);
  for (int i = 0; i < nblocks; ++i) {
    if (record) c->hook_record[i] = record[i];
    if (replay) c->hook_replay[i] = replay[i];
  }
  return 0;
}

int mi355_bresnet50_debug_tensor(const mi355_bctx* c, const char* name, void** ptr, int* dtype, int* ndim, int shape[4]) {
  MI355_ARG(c && c->arena && name && ptr && dtype && ndim && shape, "bresnet50_debug_tensor: null argument / layout-only ctx");
  const std::string n(name);
  auto set4 = [&](const void* p, int h, int w, int ch) {
    if (!p) return 1;
    *ptr = const_cast<void*>(p); *dtype = c->dtype; *ndim = 4;
    shape[0] = c->N; shape[1] = h; shape[2] = w; shape[3] = ch;
    return 0;
  };
  auto setf = [&](const float* p, int d0, int d1) {
    if (!p) return 1;
    *ptr = const_cast<float*>(p); *dtype = MI355_F32; *ndim = d1 ? 2 : 1;
    shape[0] = d0; shape[1] = d1; shape[2] = shape[3] = 0;
    return 0;
  };
  auto layer = [&](const VConv& v, const VBN& b) -> int {
    if (n == v.name + ".y") return set4(v.y, v.Hout, v.Wout, v.Coutp);
    if (n == b.name + ".out" && !b.lazy) return set4(b.out, v.Hout, v.Wout, b.Cp);
    if (n == b.name + ".save_mean") return setf(b.mean, b.Cp, 0);
    if (n == b.name + ".save_invstd") return setf(b.invstd, b.Cp, 0);
    return 1;
  };
  if (layer(c->s0, c->sb0) == 0 || layer(c->s1, c->sb1) == 0 || layer(c->s2, c->sb2) == 0) return 0;
  if (n == "stem.p") return set4(c->p, c->H / 4, c->W / 4, 64);
  for (const auto& b : c->blocks) {
    if (layer(b.c1, b.b1) == 0 || layer(b.c2, b.b2) == 0 || layer(b.c3, b.b3) == 0) return 0;
    if (b.has_ds && layer(b.ds, b.bd) == 0) return 0;
    const std::string pre = b.c1.name.substr(0, b.c1.name.size() - 5);  // strip "conv1"
    const int C4 = 4 * b.planes;
    if (n == pre + "out") return set4(b.out, b.Ho, b.Wo, C4);
    if (n == pre + "a2b" && b.stride == 2) return set4(b.a2b, b.Ho, b.Wo, b.c2.Coutp);
    if (n == pre + "scin" && b.stride == 2 && b.has_ds) return set4(b.scin, b.Ho, b.Wo, b.ds.Cinp);
    if (n == pre + "gate") return setf(b.gate, c->N, C4);
    if (n == pre + "keep" && b.scaled) return setf(b.keep, c->N, 0);
  }
  set_error("bresnet50_debug_tensor: unknown tensor '%s' (or not stored by the last forward)", name);
  return MI355_E_ARG;
}

int mi355_bresnet50_num_segments(const mi355_bctx* c) { return c ? (int)c->segs.size() : 0; }

int mi355_bresnet50_segment_range(const mi355_bctx* c, int seg, size_t* grad_begin, size_t* grad_end) {
  MI355_ARG(c && grad_begin && grad_end && seg >= 0 && seg < (int)c->segs.size(), "bresnet50_segment_range: bad arguments");
  *grad_begin = c->segs[seg].first;
  *grad_end = c->segs[seg].second;
  return 0;
}

int mi355_bresnet50_bucket_plan(const mi355_bctx* c, double bucket_cap_mb, int cap, int* n_out, size_t* begins, size_t* ends, int* last_segs) {
  MI355_ARG(c && n_out && bucket_cap_mb > 0, "bresnet50_bucket_plan: bad arguments");
  const auto bk = plan_bbuckets(c, (size_t)(bucket_cap_mb * (1 << 20) / 4));
  *n_out = (int)bk.size();
  for (int i = 0; i < (int)bk.size() && i < cap; ++i) {
    if (begins) begins[i] = bk[i].begin;
    if (ends) ends[i] = bk[i].end;
    if (last_segs) last_segs[i] = bk[i].last_seg;
  }
  return 0;
}

int mi355_bresnet50_set_comm(mi355_bctx* c, mi355_comm* comm, double bucket_cap_mb) {
  MI355_ARG(c && (comm == nullptr || bucket_cap_mb > 0), "bresnet50_set_comm: bad arguments");
  if (c->device < 0) {
    set_error("bresnet50_set_comm: layout-only ctx (created with device < 0)");
    return MI355_E_STATE;
  }
  c->comm = comm;
  c->buckets = comm ? plan_bbuckets(c, (size_t)(bucket_cap_mb * (1 << 20) / 4)) : std::vector<mi355_bctx::Bucket>();
  return 0;
}

int mi355_bresnet50_set_grad_sync(mi355_bctx* c, int on) {
  MI355_ARG(c, "bresnet50_set_grad_sync: null ctx");
  c->grad_sync = on != 0;
  return 0;
}

}  // extern "C"
