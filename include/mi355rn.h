/*
 * mi355rn.h — C-ABI of libmi355rn.so: the MI355X (gfx950) native ResNet-50 training hot path.
 *
 * This is the drop-in boundary for the path bonlime/sota_imagenet drives through torch's dispatcher:
 *   model(data)            reference call form  sota_imagenet/callbacks.py:316, built at train.py:64
 *   criterion(out, target) reference call form  sota_imagenet/callbacks.py:316, built at train.py:81
 *   loss.backward()        reference call form  sota_imagenet/callbacks.py:317
 *   optimizer.step()       reference call form  sota_imagenet/callbacks.py:309, built at train.py:92
 * The reference has no FFI of its own (it is pure Python on top of torch / cuDNN / NCCL), so every entry
 * point below cites the reference call site whose native work it replaces.
 *
 * Conventions
 *   - plain C types only: device pointers as void* / float*, sizes as int / size_t, streams as void*
 *     (a hipStream_t; pass torch.cuda.current_stream().cuda_stream).  No torch types cross the boundary.
 *   - every function returns 0 on success, a negative mi355_status otherwise; mi355_last_error() gives
 *     the message (thread-local).  Nothing throws across the boundary.
 *   - the caller (PyTorch-ROCm tensors) owns parameters / gradients / momentum / inputs / logits; the
 *     library owns only its workspace arena (saved activations, split-K partials) inside a ctx.
 *   - nothing in here synchronises the host with the device; all work is enqueued on `stream`.
 *   - there is NO CPU fallback: without a gfx950 device every compute call fails with MI355_E_HIP.
 *
 * Layouts
 *   activations  NHWC, dtype MI355_F32 or MI355_BF16           (the loader's NCHW fp32 batch,
 *                sota_imagenet/dali_dataloader.py:113-122, is converted once by mi355_ingest_*)
 *   conv weights KRSC = [Cout][KH][KW][Cin] fp32 master copy   (== a torch OIHW tensor in channels_last
 *                memory format, so state_dict() keeps torchvision names/shapes)
 *   BN / FC / loss / optimizer state: fp32.
 */
#ifndef MI355RN_H
#define MI355RN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mi355_status {
  MI355_OK = 0,
  MI355_E_ARG = -1,   /* bad argument (shape / dtype / null pointer / unsupported geometry) */
  MI355_E_HIP = -2,   /* a HIP runtime call failed (message holds file:line and hipGetErrorString) */
  MI355_E_STATE = -3, /* call order violated (e.g. backward before forward, params not bound) */
  MI355_E_NOMEM = -4  /* workspace too small / allocation failed */
} mi355_status;

/* MI355_FP8: OCP e4m3fn operand bytes — the *_fp8 entry points (their outputs are bf16) and the fp8 training step of the
 * whole-network executor (mi355_resnet50_create) */
typedef enum mi355_dtype { MI355_F32 = 0, MI355_BF16 = 1, MI355_FP8 = 2 } mi355_dtype;

/* ---- library ------------------------------------------------------------------------------------ */
const char* mi355_last_error(void);
int mi355_version(void);
/* number of HIP devices visible (0 on a GPU-less host); never fails */
int mi355_device_count(void);

/* ---- per-op entry points (unit parity tests; the executor below calls the same kernels) ----------
 * All pointers are device pointers.  `ws`/`ws_bytes` is caller-provided scratch; query the need with
 * mi355_conv2d_workspace_bytes().                                                                   */

/* scratch needed by dgrad (transposed weights) / wgrad (split-K partials) for one conv geometry */
size_t mi355_conv2d_workspace_bytes(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                    int stride, int pad);

/* y[N,Ho,Wo,Cout] = conv(x[N,H,W,Cin], w[Cout,KH,KW,Cin]); x,w,y in `dtype`.  Cin%64==0, Cout%64==0.
 * replaces cuDNN conv fwd under model(data) — callbacks.py:316 (K2 of SURVEY §2.3)                  */
int mi355_conv2d_fwd(int dtype, const void* x, const void* w, void* y, int N, int H, int W, int Cin,
                     int Cout, int KH, int KW, int stride, int pad, void* stream);

/* dx[N,H,W,Cin] = conv_transpose(dy[N,Ho,Wo,Cout], w) (+ addend[N,H,W,Cin] if non-null).
 * replaces cuDNN dgrad under loss.backward() — callbacks.py:317 (K8)                                */
/* forward conv that also leaves the BatchNorm statistics of its output as per-workgroup partial rows ([nblk][2][Cout] floats:
 * sum, sum of squares of the values AS STORED) — what the executor's convs do; *nblk = 0 when this launch shape cannot produce
 * them (then run mi355_bn_fwd_train).  partial: >= 768 * 2 * Cout floats.  Consumed by mi355_bn_fwd_train_partial.          */
int mi355_conv2d_fwd_stats(int dtype, const void* x, const void* w, void* y, float* partial, size_t partial_bytes, int* nblk,
                           int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
int mi355_conv2d_dgrad(int dtype, const void* dy, const void* w, void* dx, const void* addend, int N,
                       int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* ws,
                       size_t ws_bytes, void* stream);
/* the forward conv with the PREVIOUS layer's BatchNorm + ReLU in its operand path, as the executor's training forward launches conv2 of a
 * bottleneck (model(data), /root/reference/sota_imagenet/callbacks.py:316: BatchNorm2d -> ReLU -> Conv2d of pytorch_tools' Bottleneck; BN
 * semantics /root/reference/train.py:76, /root/reference/sota_imagenet/arg_parser.py:132):
 *   a = relu(y_in * scale[c] + shift[c]) rounded to `dtype` (what mi355_bn_apply computes from the coefficients mi355_bn_fwd_train leaves),
 *   y = conv(a, w), + the BatchNorm statistics rows of y as mi355_conv2d_fwd_stats leaves them.
 * a_out [N,H,W,Cin] and a_bits (its ReLU mask, 1 byte per 8 channels) are written by the same launch as a by-product: the weight gradient and
 * the BatchNorm backward read them; no separate bn_apply pass runs.  scale_shift: [2][Cin] floats.  bf16, 3x3 / stride 1 at the shapes the
 * generated kernels cover (MI355_E_ARG elsewhere: apply the BatchNorm with mi355_bn_apply and call mi355_conv2d_fwd_stats).                  */
int mi355_conv2d_fwd_bn_in(int dtype, const void* y_in, const float* scale_shift, const void* w, void* y, void* a_out, uint8_t* a_bits,
                           float* partial, size_t partial_bytes, int* nblk, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                           int pad, void* stream);
/* the data gradient as the executor's backward launches it (loss.backward(), /root/reference/sota_imagenet/callbacks.py:317; the
 * residual add + ReLU + BatchNorm backward autograd runs around cuDNN's dgrad there, fused into the conv's epilogue here):
 *   dx = conv_transpose(dy, w) + (addend under addend_bits — the shortcut gradient under the block output's ReLU bit mask; bits null: plain)
 * and, when `partial` is non-null, the BN-backward sums of the layer whose post-ReLU activation dx is the gradient of, as partial
 * rows [*nblk][2][Cin] floats: sum dz and sum dz * xhat, dz = dx (as stored) under bn_bits, xhat = (bn_y - bn_mean) * bn_invstd
 * (bn_y laid out like dx, masks one byte per 16-byte vector).  *nblk = 0 when the launch shape cannot produce them.
 * partial: >= 768 * 2 * Cin floats.  dx may alias addend.
 * addend_sub2 = 1: `addend` is [N][H/2][W/2][Cin] and stands for the full-resolution tensor that is zero at odd rows / columns — the
 * data gradient of the stride-2 1x1 downsample convolution of a stage's first block, which the executor then never writes at full size
 * (bf16, 1x1 / stride 1 launches with even H and W; MI355_E_ARG elsewhere).                                                         */
int mi355_conv2d_dgrad_bn(int dtype, const void* dy, const void* w, void* dx, const void* addend, const uint8_t* addend_bits, int addend_sub2,
                          const void* bn_y, const uint8_t* bn_bits, const float* bn_mean, const float* bn_invstd, float* partial,
                          size_t partial_bytes, int* nblk, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          void* ws, size_t ws_bytes, void* stream);
/* the same with the sums under a LEAKY ReLU mask: dz = dx where the bit is set, dx * slope elsewhere — the activation of BASELINE configs[3]'s model
 * (`norm_act: leaky_relu`, /root/reference/configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51; autograd's LeakyReLU + BatchNorm backward
 * around cuDNN's dgrad under loss.backward(), /root/reference/sota_imagenet/callbacks.py:317).  slope = 0: mi355_conv2d_dgrad_bn.  slope = 0.01 at the
 * shapes of that model's bottlenecks (bf16, generated kernels with that epilogue); any other slope or shape: MI355_E_ARG (run the BatchNorm backward's
 * own reduction, mi355_bn_bwd).                                                                                                                    */
int mi355_conv2d_dgrad_bn_leaky(int dtype, const void* dy, const void* w, void* dx, const void* addend, const uint8_t* addend_bits, int addend_sub2,
                                const void* bn_y, const uint8_t* bn_bits, const float* bn_mean, const float* bn_invstd, float slope, float* partial,
                                size_t partial_bytes, int* nblk, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                void* ws, size_t ws_bytes, void* stream);

/* dw[Cout,KH,KW,Cin] (fp32) = sum_{n,oh,ow} dy (x) x.  beta=0 overwrites, beta=1 accumulates.
 * replaces cuDNN wgrad under loss.backward() — callbacks.py:317 (K8)                                */
int mi355_conv2d_wgrad(int dtype, const void* dy, const void* x, float* dw, float beta, int N, int H,
                       int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* ws,
                       size_t ws_bytes, void* stream);

/* real-image ingest: the GPU half of the reference's DALI pipelines, after the JPEG decoder —
 * train sota_imagenet/dali_dataloader.py:69-83 (random crop at decode, fn.resize size=S INTERP_TRIANGULAR, or INTERP_CUBIC by
 * coin), :85-114 (gaussian blur window 11, colour twist, grey via hsv saturation, random erasing with the mean), :116-125
 * (crop_mirror_normalize: mirror coin, mean 127.5 / std 51 (:27-29), FLOAT, NCHW); val :144-157 (resize_shorter, centre crop).
 * `packed` (device) holds the decoded u8 RGB crops of one batch back to back (HWC, tight rows); sample n is described by
 * crops[n]: resize its h x w rectangle to rh x rw (filter 0: triangular = antialiased bilinear, 1: a = -0.5 cubic), take the
 * S x S window at (oy, ox), [augment[n]: gaussian blur (sigma > 0), 3x4 colour matrix on 0..255 values + clamp, luma if gray,
 * up to 4 rectangles y0,x0,y1,x1 (window coordinates before the mirror) filled with the mean], mirror if asked, write
 * (v - mean) / std to out_nchw[n] (fp32 [N,3,S,S], what mi355_resnet50_forward takes).  Tables are passed twice: the host copy
 * is validated against packed_bytes before the launch (MI355_E_ARG on a rectangle that leaves the buffer or a window that
 * leaves the resized image), the device copy is what the kernels read.  `scratch` (N*3*S*S floats) is needed only when some
 * sample of the batch has blur_sigma > 0 (two launches then), else NULL.                                                */
typedef struct mi355_crop {
  unsigned long long offset;
  int h, w, rh, rw, oy, ox, mirror, filter;
} mi355_crop;
typedef struct mi355_augment {
  float color[12];
  float blur_sigma;
  int gray, nbox, pad;
  int box[4][4];
} mi355_augment;
int mi355_ingest_u8(const unsigned char* packed, size_t packed_bytes, const mi355_crop* crops_host, const mi355_crop* crops_dev,
                    int N, int S, float mean, float std, float* out_nchw, void* stream);
int mi355_ingest_u8_aug(const unsigned char* packed, size_t packed_bytes, const mi355_crop* crops_host,
                        const mi355_crop* crops_dev, const mi355_augment* aug_host, const mi355_augment* aug_dev, int N, int S,
                        float mean, float std, float* scratch, float* out_nchw, void* stream);

/* fp8 (OCP e4m3fn) operand path of the convolution — BASELINE.json configs[4] "fp8 MFMA convs" (the reference has no fp8
 * path; the conv calls being replaced are the same as above, callbacks.py:316-317).  Per-tensor scaling:
 *   q = mi355_quantize_fp8(x * scale)  (saturating, round to nearest even; n a multiple of 8; source f32 or bf16)
 *   y_bf16 = conv(xq, wq) * oscale,  oscale = 1 / (scale_x * scale_w), fp32 accumulation on v_mfma_f32_16x16x32_fp8_fp8.
 * Layouts as above with 1-byte elements (x NHWC, w KRSC; dgrad takes the weights already transposed to [Cin][KH][KW][Cout]);
 * Cin and Cout multiples of 128, output width >= 2 (MI355_E_ARG otherwise: there is no second fp8 kernel to fall back on).
 * Same 8-wave kernel as the bf16 path with half the operand bytes (conv_igemm8.hip).   */
int mi355_quantize_fp8(int src_dtype, const void* x, void* q, float scale, size_t n, void* stream);
int mi355_conv2d_fwd_fp8(const void* xq, const void* wq, void* y, float oscale, int N, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, void* stream);
int mi355_conv2d_dgrad_fp8(const void* dyq, const void* wtq, void* dx, float oscale, int N, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride, int pad, void* stream);
/* dw fp32 KRSC = beta * dw + oscale * sum_pixels dyq (x) xq over e4m3 operands (v_mfma_f32_32x32x16_fp8_fp8; both operands are read
 * TRANSPOSED out of LDS with ds_read_b64_tr_b8 — the reduction index, the pixel, is the slow index of both NHWC tensors).
 * ws: mi355_conv2d_workspace_bytes(MI355_BF16, ...) + 256 bytes.                                                                   */
int mi355_conv2d_wgrad_fp8(const void* dyq, const void* xq, float* dw, float beta, float oscale, int N, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream);

/* the 7x7/2 stem on the loader's NCHW fp32 batch: ingest (NCHW fp32 -> zero-padded NHWC4 `dtype`),
 * forward y[N,H/2,W/2,64], and wgrad dw[64,7,7,3] fp32.  xpad is scratch of
 * mi355_stem_xpad_bytes() bytes that must be ZEROED once by the caller before the first ingest.
 * replaces DALI's NCHW hand-off + cuDNN stem conv — dali_dataloader.py:113-122, callbacks.py:316 (K1) */
size_t mi355_stem_xpad_bytes(int dtype, int N, int H, int W);
int mi355_stem_ingest(int dtype, const float* x_nchw, void* xpad, int N, int H, int W, void* stream);
int mi355_stem_fwd(int dtype, const void* xpad, const float* w_krsc, void* y, int N, int H, int W,
                   void* ws, size_t ws_bytes, void* stream);
int mi355_stem_wgrad(int dtype, const void* dy, const void* xpad, float* dw, float beta, int N, int H,
                     int W, void* ws, size_t ws_bytes, void* stream);
size_t mi355_stem_workspace_bytes(int dtype, int N, int H, int W);

/* BatchNorm2d, training mode, over x[M,C] (M = N*H*W rows, NHWC):
 *   mean/var over M (biased var to normalise, unbiased into running_var), eps inside the sqrt,
 *   running = (1-momentum)*running + momentum*batch.   Outputs save_mean/save_invstd [C] for backward.
 *   out = act(x_hat*gamma + beta (+ residual)),   act by `relu`: 0 identity, 1 ReLU, 2 leaky ReLU (slope 0.01 — the
 *   `norm_act: leaky_relu` of the BResNet-50 configs, BResNet50_encoder.yaml:49-50; same codes in mi355_bn_bwd).
 *   ws: >= mi355_bn_workspace_bytes(C) bytes.
 * replaces cuDNN BatchNormalizationForwardTraining + ATen relu/add — callbacks.py:316 (K3,K4);
 * momentum semantics: train.py:76 / arg_parser.py:132                                              */
size_t mi355_bn_workspace_bytes(int C);
int mi355_bn_fwd_train(int dtype, const void* x, const void* residual, void* out, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float* save_mean,
                       float* save_invstd, int M, int C, float eps, float momentum, int relu, void* ws,
                       size_t ws_bytes, void* stream);
/* inference mode: uses running stats */
/* the same from partial rows a conv epilogue left (mi355_conv2d_fwd_stats): no pass over x for the statistics; ws: 2*C floats */
int mi355_bn_fwd_train_partial(int dtype, const void* x, const void* residual, void* out, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float* save_mean,
                               float* save_invstd, int M, int C, float eps, float momentum, int relu, const float* partial,
                               int nblk, void* ws, size_t ws_bytes, void* stream);
int mi355_bn_fwd_eval(int dtype, const void* x, const void* residual, void* out, const float* gamma,
                      const float* beta, const float* running_mean, const float* running_var, int M,
                      int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream);
/* backward of out = act(bn(x) (+residual)):  dz = dout * [out>0] (if relu),  dx = bn_bwd(dz),
 * dgamma/dbeta fp32 (beta_acc=0 overwrite / 1 accumulate).  If dz_out != NULL the masked gradient is
 * also stored (it is the residual branch's gradient).
 * replaces cuDNN BN bwd + ATen threshold_backward — callbacks.py:317 (K8)                           */
int mi355_bn_bwd(int dtype, const void* dout, const void* out, const void* x, const float* gamma,
                 const float* save_mean, const float* save_invstd, void* dx, void* dz_out,
                 float* dgamma, float* dbeta, float beta_acc, int M, int C, int relu, void* ws,
                 size_t ws_bytes, void* stream);

/* MaxPool 3x3 stride 2 pad 1 on NHWC; idx[N,Ho,Wo,C] uint8 = window position of the first maximum.
 * replaces ATen max_pool2d_with_indices fwd/bwd — callbacks.py:316-317 (K5)                         */
int mi355_maxpool_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C,
                      void* stream);
int mi355_maxpool_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W,
                      int C, void* stream);

/* global average pool x[N,HW,C] -> pooled[N,C] fp32, and its backward (broadcast of dpooled/HW).
 * replaces ATen mean / its backward — callbacks.py:316-317 (K6)                                     */
int mi355_gap_fwd(int dtype, const void* x, float* pooled, int N, int HW, int C, void* stream);
int mi355_gap_bwd(int dtype, const float* dpooled, void* dx, int N, int HW, int C, void* stream);

/* label-smoothed softmax cross-entropy on float (one-hot or soft) targets, reduction = mean:
 *   loss = mean_n[ (1-s) * -(sum_c y*logp) + s * -(mean_c logp) ],  logp = log_softmax(logits)
 *   dlogits = d loss / d logits (already divided by N) times grad_scale.
 * loss: 1 float on device.  row_loss: N floats of scratch.  dlogits may be NULL (evaluation).
 * replaces pytorch_tools.losses.smooth.CrossEntropyLoss — arg_parser.py:140-142, callbacks.py:316 (K7)*/
int mi355_ce_loss(const float* logits, const float* target, float smoothing, float grad_scale,
                  float* loss, float* row_loss, float* dlogits, int N, int C, void* stream);

/* fused SGD with momentum on a flat fp32 range (torch.optim.SGD semantics, dampening 0, no nesterov):
 *   g = g*grad_scale + wd*p;  m = mu*m + g;  p -= lr*m       (m must start at 0 => first step m = g)
 * replaces torch.optim._multi_tensor.SGD.step — arg_parser.py:136-138, callbacks.py:309 (K10)       */
int mi355_sgd_step(float* p, const float* g, float* m, size_t n, float lr, float momentum,
                   float weight_decay, float grad_scale, void* stream);
/* the same step + the exponential moving average of the updated parameters in the same pass:
 *   ema += (1 - ema_decay) * (p_new - ema)      (= ema_decay * ema + (1 - ema_decay) * p_new)
 * replaces pytorch_tools' ModelEma callback over the parameters (train.py:111-112, `ema_decay` of
 * configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:59) when the optimizer steps every batch */
int mi355_sgd_step_ema(float* p, const float* g, float* m, float* ema, size_t n, float lr, float momentum,
                       float weight_decay, float grad_scale, float ema_decay, void* stream);

/* ---- BResNet-50 variant blocks (BASELINE configs[3]) ---------------------------------------------------------------
 * The reference builds that model as pytorch_tools.models.resnet50(stem_type="deep", antialias=True, attn_type="eca",
 * norm_layer="inplaceabn", norm_act="leaky_relu", drop_rate=0.2, drop_connect_rate=0.2) —
 * configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51 — and wraps every conv in weight standardisation
 * (train.py:66-67).  Convolutions and BN reuse the entry points above; these are the additional ops, all NHWC.
 *   blurpool    3x3 binomial [1,2,1]x[1,2,1]/16, stride 2, reflect padding 1 (anti-aliased down-sampling); H, W even
 *   avgpool2    2x2 average, stride 2 (the anti-aliased shortcut of a stride-2 block)
 *   maxpool3s1  3x3 max, stride 1, pad 1 + u8 argmax (first maximum in window scan order): the anti-aliased stem pool
 *   eca         y = x * sigmoid(conv1d_k(GAP(x)))[n][c]; k odd <= 9, zero padded over the channel axis.  pooled / gate
 *               [N][C] fp32 are outputs kept for backward; eca_bwd returns dx and the k conv-weight gradients
 *               (beta = 0 overwrite / 1 accumulate), ws = 2*N*C + 1152 floats of scratch
 *   weight_std  w_hat[o] = (w[o] - mean_o) * rsqrt(var_o + eps) over the K = KH*KW*Cin weights of output channel o
 *               (biased variance) and its backward dw = invstd * (g - mean(g) - w_hat * mean(g * w_hat))
 *   residual_act  out = act(branch * scale_n[n] + shortcut): drop-connect keep/scale per sample (scale_n may be NULL),
 *               shortcut add (may be NULL), activation code as in mi355_bn_fwd_train; backward from `out`
 *   keep_scale  keep[i] = u_i >= p ? 1/(1-p) : 0 from a counter-based generator (seed, counter): the drop-connect
 *               sample scales and, multiplied onto the pooled features (mi355_mul_f32), dropout                     */
int mi355_blurpool_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);
int mi355_blurpool_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int mi355_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);
int mi355_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int mi355_maxpool3s1_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int mi355_maxpool3s1_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                         void* stream);
int mi355_eca_fwd(int dtype, const void* x, const float* w, int k, void* y, float* pooled, float* gate, int N,
                  int HW, int C, void* stream);
int mi355_eca_bwd(int dtype, const void* dy, const void* x, const float* w, int k, const float* pooled,
                  const float* gate, void* dx, float* dw, float beta, float* ws, int N, int HW, int C, void* stream);
int mi355_weight_std_fwd(const float* w, float* w_hat, float* mean, float* invstd, int Cout, int K, float eps,
                         void* stream);
int mi355_weight_std_bwd(const float* dw_hat, const float* w_hat, const float* invstd, float* dw, float beta,
                         int Cout, int K, void* stream);
int mi355_residual_act_fwd(int dtype, const void* branch, const float* scale_n, const void* shortcut, void* out,
                           int N, size_t HWC, int act, void* stream);
int mi355_residual_act_bwd(int dtype, const void* dout, const void* out, const float* scale_n, void* dbranch,
                           void* dshortcut, int N, size_t HWC, int act, void* stream);
int mi355_keep_scale(float* keep, size_t n, float p, unsigned long long seed, unsigned long long counter,
                     void* stream);
int mi355_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream);

/* Mixup / CutMix with the previous batch, on the device (CutmixMixup — sota_imagenet/callbacks.py:232-247; the
 * pytorch_tools Cutmix / Mixup bases it combines mix the batch with the PREVIOUS one under a random permutation).
 *   mi355_mix_sample  draws one batch's decisions on the device from (seed, counter): apply at all (probability `prob`),
 *                     CutMix or Mixup (coin, callbacks.py:242; `allow`: 1 Mixup only, 2 CutMix only, 3 both, 0 neither),
 *                     lambda ~ Beta(alpha, alpha), the CutMix box, a permutation of the N <= 1024 samples; the result
 *                     stays in `params` (mi355_mix_params_bytes(N) bytes of device memory: {int mode 0/1/2, float lambda,
 *                     int y1, y2, x1, x2, float box_area_fraction, int pad, int perm[N]}) — no host round trip.
 *   mi355_mix_apply   out / tout <- data[N,C,H,W] fp32 (the loader's NCHW batch) and target[N,classes] fp32 soft targets
 *                     mixed with prev_in / tprev_in under perm (out may alias data, tout target: in place); the UNMIXED
 *                     batch is stored to prev_out / tprev_out for the next step (double-buffered: pass the two buffers
 *                     in alternating roles).
 *                     Mixup: lambda*x + (1-lambda)*prev[perm]; CutMix: the box is pasted from prev[perm], the target
 *                     weights are the real box area.  W and classes must be multiples of 4.                         */
size_t mi355_mix_params_bytes(int N);
int mi355_mix_sample(void* params, unsigned long long seed, unsigned long long counter, int N, int H, int W,
                     float cutmix_alpha, float mixup_alpha, float prob, int allow, void* stream);
int mi355_mix_apply(const float* data, float* out, const float* prev_in, float* prev_out, const float* target,
                    float* tout, const float* tprev_in, float* tprev_out, const void* params, int N, int C, int H,
                    int W, int num_classes, void* stream);

/* ---- whole-network executor: torchvision-layout ResNet-50 v1.5 ----------------------------------
 * replaces hydra.utils.call(cfg.model) -> pytorch_tools.models.resnet50 (train.py:64,
 * configs/hydra_exp/1.r50_baseline.yaml:22-23) and everything autograd runs beneath it.            */
typedef struct mi355_ctx mi355_ctx;

/* N = per-GPU batch, H = W = image size (multiple of 32), dtype = activation/compute dtype of the convs
 * (accumulation, BN statistics, FC, loss and optimizer state are always fp32).
 * device < 0 creates a LAYOUT-ONLY ctx (no HIP call, no memory): the tensor table, flat sizes, segment
 * ranges and FLOP counts can be queried on a GPU-less host; bind/forward/backward fail with E_STATE.  */
int mi355_resnet50_create(mi355_ctx** out, int device, int dtype, int N, int H, int W, int num_classes);
int mi355_resnet50_destroy(mi355_ctx* ctx);

/* Parameter / buffer table (torchvision names).  Flat layouts are in REVERSE execution order (fc
 * first, stem last) so that gradient buckets complete front-to-back during backward.
 *   kind: 0 = parameter (offset into flat params/grads), 1 = buffer (offset into flat buffers:
 *         running_mean / running_var; num_batches_tracked is kept host-side by the caller).
 *   shape: torch logical shape (conv: [Cout,Cin,KH,KW], stored channels_last = KRSC).               */
int mi355_resnet50_num_tensors(const mi355_ctx* ctx);
int mi355_resnet50_tensor_info(const mi355_ctx* ctx, int idx, char* name, int name_cap, int* kind,
                               size_t* offset, int* ndim, int shape[4]);
size_t mi355_resnet50_flat_param_elems(const mi355_ctx* ctx);  /* incl. alignment / FC row padding */
size_t mi355_resnet50_flat_buffer_elems(const mi355_ctx* ctx);
size_t mi355_resnet50_workspace_bytes(const mi355_ctx* ctx);

/* caller-owned flat fp32 device arrays; must outlive the ctx.  Padding elements must be zero. */
int mi355_resnet50_bind(mi355_ctx* ctx, float* params, float* grads, float* buffers);

/* logits[N,num_classes] fp32 = model(x_nchw[N,3,H,W] fp32).  training!=0: batch statistics, running
 * stats updated with `bn_momentum`, activations saved for backward.  training==0: running stats.
 * Streams: the dependent kernel chain is issued to `stream`; independent work (weight gradients, the
 * downsample branch) goes to a side stream the ctx owns, forked from / joined to `stream` with events, so
 * on return everything is ordered on `stream` as if it had run there (MI355_WGRAD_STREAM=0 in the
 * environment at create time keeps every kernel on `stream`).  Results are bit-identical either way.
 * fp32 contexts cut the tiles of a partial last round along K across workgroups (stream-K, bounded spins).  Should a
 * hand-off ever time out, the kernel raises an error word that is copied to pinned host memory behind the kernels
 * of the call (no host wait): forward / backward / profile_read of that context then fail with MI355_E_STATE from
 * the first call that finds it set — at the latest the call after the next host-device synchronisation.          */
int mi355_resnet50_forward(mi355_ctx* ctx, const float* x_nchw, float* logits, int training,
                           float bn_momentum, void* stream);

/* Backward is split into mi355_resnet50_num_segments() segments (0 = fc, then one per bottleneck block
 * from layer4.2 down to layer1.0, last = stem) so the caller can launch a gradient all-reduce on a side
 * stream as soon as a segment's slice [grad_begin, grad_end) of the flat gradient array is complete.
 * Segments must be run in order 0..n-1 after a training forward.  Gradients OVERWRITE the flat array
 * (accumulate != 0: add into it, for accumulate_steps > 1 — arg_parser.py:85-86).  Every call ends by
 * joining the side stream: prefer one call per gradient bucket over one call per segment.            */
int mi355_resnet50_num_segments(const mi355_ctx* ctx);
int mi355_resnet50_segment_range(const mi355_ctx* ctx, int seg, size_t* grad_begin, size_t* grad_end);
int mi355_resnet50_backward(mi355_ctx* ctx, const float* dlogits, int seg_begin, int seg_end,
                            int accumulate, void* stream);


/* ---- BResNet-50 (BASELINE configs[3]) as a static executor — csrc/bresnet_exec.cpp -------------------------------------
 * The model the reference builds with `_target_: pytorch_tools.models.resnet50` and the model_params of
 * configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51 (deep stem, anti-aliasing, ECA, leaky-ReLU ABN, drop-connect,
 * dropout) + weight standardisation of every conv (train.py:66-67): one call per forward (train.py:64 `model(data)` under
 * callbacks.py:316) and one per backward (callbacks.py:317).  Same protocol as mi355_resnet50_*: create (device < 0: layout-only,
 * works without a GPU), tensor table (pytorch_tools names; kind 0 parameter / 1 buffer; conv weights logical [Cout,Cin,KH,KW] over
 * [Cout][KH][KW][Cin] memory; FC rows padded to a multiple of 128 inside the flat array), bind the three flat fp32 arrays, run.
 *   forward   training != 0: batch statistics, running stats updated with `bn_momentum`, everything backward needs is kept.
 *             Drop-connect / dropout (training only, rates from mi355_bresnet50_set_drop): block i > 0 scales its branch by
 *             keep_scale(N, rate * i / 16, seed, step * 64 + i), the pooled features by keep_scale(N * 2048, drop_rate, seed,
 *             step * 64 + 63) (mi355_keep_scale).  keep_override != NULL replaces the generator: 16 pointers, entry i = the [N]
 *             sample scales of block i or NULL (none); dropout_override = [N][2048] scales or NULL (none) — the test hook.
 *   backward  of the last training forward; accumulate != 0 adds to the flat gradient array.                                   */
typedef struct mi355_bctx mi355_bctx;
int mi355_bresnet50_create(mi355_bctx** out, int device, int dtype, int N, int H, int W, int num_classes, int weight_std);
int mi355_bresnet50_destroy(mi355_bctx* ctx);
int mi355_bresnet50_num_tensors(const mi355_bctx* ctx);
int mi355_bresnet50_tensor_info(const mi355_bctx* ctx, int idx, char* name, int name_cap, int* kind, size_t* offset, int* ndim,
                                int* shape);
size_t mi355_bresnet50_flat_param_elems(const mi355_bctx* ctx);
size_t mi355_bresnet50_flat_buffer_elems(const mi355_bctx* ctx);
size_t mi355_bresnet50_workspace_bytes(const mi355_bctx* ctx);
int mi355_bresnet50_bind(mi355_bctx* ctx, float* params, float* grads, float* buffers);
int mi355_bresnet50_set_drop(mi355_bctx* ctx, float drop_rate, float drop_connect_rate, unsigned long long seed);
int mi355_bresnet50_forward(mi355_bctx* ctx, const float* x_nchw, float* logits, int training, float bn_momentum,
                            unsigned long long step, const float* const* keep_override, const float* dropout_override,
                            void* stream);
int mi355_bresnet50_backward(mi355_bctx* ctx, const float* dlogits, int accumulate, void* stream);
int mi355_bresnet50_flops(const mi355_bctx* ctx, double* fwd_flops, double* train_flops);
/* Data parallelism of this executor (reference: DistributedDataParallel at /root/reference/train.py:113-114 around the configs[3] model):
 * the backward call runs mi355_bresnet50_num_segments() segments — 0 = the head, then the bottlenecks last to first, then the stem;
 * the flat gradient array is laid out in FORWARD (pytorch_tools registration) order, so the segments descend through it — and, with a
 * communicator attached, issues ONE mean all-reduce per bucket of consecutive segments (>= bucket_cap_mb MiB, the last bucket cut once
 * more: the rule of mi355_resnet50_set_comm) on the communicator's stream as soon as the bucket's last segment has been enqueued on
 * both streams; the caller's stream waits for the last one before the call returns.  set_grad_sync(0): no collective (DDP.no_sync()). */
typedef struct mi355_comm mi355_comm;
int mi355_bresnet50_num_segments(const mi355_bctx* ctx);
int mi355_bresnet50_segment_range(const mi355_bctx* ctx, int seg, size_t* grad_begin, size_t* grad_end);
int mi355_bresnet50_bucket_plan(const mi355_bctx* ctx, double bucket_cap_mb, int cap, int* n_out, size_t* begins, size_t* ends, int* last_segs);
int mi355_bresnet50_set_comm(mi355_bctx* ctx, mi355_comm* comm, double bucket_cap_mb);
int mi355_bresnet50_set_grad_sync(mi355_bctx* ctx, int on);

/* Test hook of the BResNet-50 executor (as mi355_resnet50_debug_tensor): device pointer / shape of a tensor of the last forward, by
 * name: "<conv>.y" (conv output, channels padded to a multiple of 64), "<bn>.out" (where the last forward stored it: bn3 / the
 * downsample BN are applied inside the fused ECA pass by default and have none), "<bn>.save_mean|save_invstd", "stem.p" (the
 * anti-aliased pool's output = layer1.0's input), "<block>.out|gate|keep|a2b|scin" (block output, ECA gate [N][C], drop-connect
 * scales [N] when sampled, blur-pooled a2 / average-pooled input of a striding block).  Read-only use.                             */
int mi355_bresnet50_debug_tensor(const mi355_bctx* ctx, const char* name, void** ptr, int* dtype, int* ndim, int shape[4]);
/* test hook of the NEXT mi355_bresnet50_backward call (cleared by it): for block i (0 = layer1.0 ... 15 = layer4.2), record[i] (device memory, the
 * size of that block's output in the compute dtype, or null) receives the gradient wrt the block's output the moment the block's backward starts,
 * and replay[i] (or null) overwrites that gradient first — teacher forcing between backward segments: two builds / switch settings fed the SAME
 * incoming gradient per block must agree per segment to rounding, where free-running backward passes amplify a last-bit difference (the
 * counterpart of mi355_resnet50_force_grad; the reference has no such hook: autograd under loss.backward(), /root/reference/sota_imagenet/callbacks.py:317). */
int mi355_bresnet50_grad_hooks(mi355_bctx* ctx, void* const* record, const void* const* replay, int nblocks);

/* ---- gradient collective inside the boundary: RCCL over xGMI, one process per GPU ---------------------------------
 * replaces torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank]) — train.py:113-114, process group
 * train.py:58-61 — i.e. the one-time rank-0 broadcast of parameters / buffers and the bucketed gradient MEAN all-reduce
 * overlapped with backward.  RCCL is bound at run time (librccl.so.1); mi355_comm_available() says whether it was found.
 *   bootstrap  rank 0 calls mi355_comm_unique_id(id) (MI355_COMM_ID_BYTES bytes) and hands `id` to the other ranks through
 *              the launcher's own channel (the reference has one: init_process_group("nccl", "env://"), train.py:61);
 *              every rank then calls mi355_comm_create(&comm, id, nranks, rank, device) (collective: ncclCommInitRank).
 *   data path  mi355_resnet50_set_comm(ctx, comm, bucket_cap_mb): consecutive backward segments form buckets of at least
 *              bucket_cap_mb MiB of the flat gradient array (the last bucket is cut once more: its trailing segments up to
 *              bucket_cap_mb / 8 — stem, layer 1, the end of layer 2 — form a small bucket of their own, so that only a few MB
 *              are reduced after backward has ended); mi355_resnet50_backward then launches, when a bucket's last
 *              segment has been enqueued, ONE mean all-reduce over the bucket's contiguous slice on a HIP stream the
 *              communicator owns, ordered by events behind the kernels that produce it (both executor streams), and makes
 *              `stream` wait for the last one before it returns — so ONE backward call covers all segments, the collective
 *              overlaps the rest of backward, and the caller's next kernel (the optimizer step) sees reduced gradients.
 *              No host thread, no host wait.  comm == NULL detaches.  mi355_resnet50_bucket_plan reports the buckets a
 *              cap would produce (works on a layout-only ctx; n_out = number of buckets, arrays filled up to `cap`).
 *   xGMI is point-to-point (7 links per GPU, ring collectives are per-link bound): keep buckets few and large.      */
#define MI355_COMM_ID_BYTES 128
/* (mi355_comm: declared above, with the BResNet-50 executor's data-parallel entry points) */
int mi355_comm_available(void);
int mi355_comm_unique_id(void* id_out);
int mi355_comm_create(mi355_comm** out, const void* id, int nranks, int rank, int device);
int mi355_comm_destroy(mi355_comm* comm);
int mi355_comm_nranks(const mi355_comm* comm);
/* in-place collectives on fp32 device arrays, enqueued on `stream` (construction-time broadcast of rank `root`'s
 * parameters / BN buffers; a plain mean all-reduce for callers that keep their own schedule) */
int mi355_comm_broadcast(mi355_comm* comm, float* buf, size_t n, int root, void* stream);
int mi355_comm_allreduce_mean(mi355_comm* comm, float* buf, size_t n, void* stream);
int mi355_resnet50_set_comm(mi355_ctx* ctx, mi355_comm* comm, double bucket_cap_mb);
/* CUs this library leaves to the communication library's kernels while a step runs (RCCL's all-reduce kernels need CUs of their own;
 * the reference leaves that to NCCL and the CUDA scheduler, /root/reference/train.py:113-114): every grid sized from the CU count plans
 * for CUs - n.  n: a non-negative multiple of 8 (the same number per XCD).  Process-global; call before creating contexts.        */
int mi355_set_reserved_cus(int n);
/* the per-launch tile knobs (MI355_IGEMM8, MI355_IGEMM_BIG, MI355_STEM_DIRECT, MI355_STEM_TH, MI355_STEM_DBG: test hooks / A/B switches)
 * are read from the environment once; this re-reads them (tests flip them between launches of one process)                          */
int mi355_reload_knobs(void);
/* name of the kernel the calling thread's last convolution / weight-gradient launch went to: the symbol of a generated gfx950 kernel
 * (dconv_l3_s1, po_k256_b256_s2_a2, wg3_l2 ...) or the implicit-GEMM tile family (igemm<bf16,128,128,2> ...).  A debugging query: the
 * tests pin the selection rules with it (cuDNN's own algorithm choice under /root/reference/sota_imagenet/callbacks.py:316-317 is not
 * observable in the reference).  The string stays valid until the thread's next launch.                                            */
const char* mi355_last_conv_kernel(void);
/* measurement stand-in for a collective's CU footprint on one GPU: `workgroups` 256-thread workgroups (16 KiB of LDS each) that hold
 * their CU slots for `usec` microseconds without memory traffic (tools/reserve_cus_ab.py; not part of the training path)           */
int mi355_comm_standin(int workgroups, int usec, void* stream);
/* DistributedDataParallel.no_sync() (gradient accumulation, accumulate_steps > 1 — arg_parser.py:85-86, the Runner's inner step):
 * on == 0 makes the following backward calls of this ctx skip the bucket all-reduces (gradients stay rank-local and keep
 * accumulating); the last micro-step runs with on != 0 (the default) and reduces the accumulated sums once.        */
int mi355_resnet50_set_grad_sync(mi355_ctx* ctx, int on);
/* Record of every collective issued through `comm` since creation / the last reset, in issue order (at most 4096 kept):
 * kinds[i] 0 = bucket mean all-reduce issued by mi355_resnet50_backward over [begins[i], ends[i]) of the flat gradient array
 * (elements), 1 = mi355_comm_broadcast of ends[i] elements, 2 = mi355_comm_allreduce_mean of ends[i] elements.
 * n_out = number of records (arrays are filled up to `cap`); reset != 0 clears the record afterwards.  Host-side only —
 * lets a test assert that one backward reduced every gradient element exactly once, in bucket order.              */
int mi355_comm_stats(mi355_comm* comm, int reset, int cap, int* n_out, int* kinds, size_t* begins, size_t* ends);
int mi355_resnet50_bucket_plan(const mi355_ctx* ctx, double bucket_cap_mb, int cap, int* n_out, size_t* begins,
                               size_t* ends, int* last_segs);

/* dtype MI355_FP8 in mi355_resnet50_create = BASELINE.json configs[4] "fp8 MFMA convs": every tensor stays bf16 (fp32 accumulation,
 * statistics, master weights as in the bf16 step) and the forward / data-gradient convolutions of every layer whose channel counts
 * are multiples of 128 on both sides (layers 2-4) read OCP e4m3 twins of their operands through v_mfma_f32_16x16x32_fp8_fp8; the
 * weight gradients, the stem, layer 1 and the FC stay on bf16 operands.  No standalone quantise pass exists: the BN-apply /
 * BN-backward-apply kernels that produce an activation / gradient tensor write its twin in the same pass (q = e4m3(bf16(v) *
 * scale), saturating) and the weight preparation writes the weights' twins.  Scaling is per tensor and DELAYED: a producer also
 * records max|v| (atomicMax), and scale(step k+1) = 448 / (2 * amax(step k)).  The first training step of a ctx has no amaxes
 * yet: it runs bf16 operands and only records them (forward and backward); from the 2nd step on the convs read the twins.
 * Inference forwards always read the bf16 tensors.  mi355_resnet50_fp8_state: whether the last training step's
 * forward / backward convs read the twins, and how many layers do.  Conv call sites replaced: callbacks.py:316-317; stage
 * schema this dtype is used with: arg_parser.py:65-72, dali_dataloader.py:213-239.
 * debug_tensor names of an fp8 ctx: "<conv>.xq" / ".wq" / ".wtq" (e4m3 bytes, dtype MI355_FP8), "<conv>.sx" / ".sw" / ".sdy" (scales). */
int mi355_resnet50_fp8_state(const mi355_ctx* ctx, int* fwd_on, int* bwd_on, int* n_fwd_layers, int* n_dgrad_layers);

/* algorithmic work of the ctx's conv/FC kernels (2 FLOP/MAC, padding-free): forward and fwd+bwd */
int mi355_resnet50_flops(const mi355_ctx* ctx, double* fwd_flops, double* train_flops);
/* which kernel each convolution of the network went to in its last forward / data-gradient / weight-gradient launch, one text line
 * per layer: "<conv name> fwd=<kernel> dgrad=<kernel> wgrad=<kernel>" ('-' = not launched yet; names as mi355_last_conv_kernel).
 * Writes at most cap bytes (NUL-terminated) and the size a complete table needs.  A debugging query: the tests assert the selection
 * rules of the bs-256 plan with it (the reference's counterpart, cuDNN's algorithm choice under
 * /root/reference/sota_imagenet/callbacks.py:316-317, is not observable).                                                         */
int mi355_resnet50_kernel_table(const mi355_ctx* ctx, char* out, size_t cap, size_t* needed);

/* Test hook: device pointer / shape of an internal tensor of the last step, by name:
 *   "<conv>.y" raw conv output, "<block>.a1|a2|out" post-activation tensors (e.g. "layer1.0.out"),
 *   "<bn>.save_mean|save_invstd", "stem.p0" (the stem's BN+ReLU+maxpool output; the 112x112 activation itself
 *   is never stored), "pooled", "dpooled", "gG0".."gG1" (block-output gradients of the last backward),
 *   "grad.cur" (between two backward segments: the gradient the next segment starts from).
 * shape is NHWC (ndim 4) or [n] / [N,C]; dtype is MI355_F32 or the ctx dtype.  Read-only use.     */
int mi355_resnet50_debug_tensor(const mi355_ctx* ctx, const char* name, void** ptr, int* dtype, int* ndim,
                                int shape[4]);
/* Test hook (teacher forcing): between two calls of mi355_resnet50_backward() that split the segments, replace the gradient the next
 * segment starts from ("grad.cur": wrt that block's output, or wrt the stem's pooled output before the last segment) by `g` (device
 * memory, the ctx dtype, exactly the tensor's bytes).  The BN-backward sums the producing kernel's epilogue left for that tensor are
 * dropped (the next segment re-reduces them from g).  Two kernel selections fed the SAME incoming gradient per segment differ by
 * that segment's own kernels only: what autograd's per-node gradcheck does for the graph under loss.backward()
 * (/root/reference/sota_imagenet/callbacks.py:317).  MI355_E_STATE outside a split backward.                                       */
int mi355_resnet50_force_grad(mi355_ctx* ctx, const void* g, size_t bytes, void* stream);

/* HIP-event timing of kernel classes inside a step, recorded on the stream the kernels are launched on.
 * mi355_resnet50_profile(ctx, class_mask): bit k set => every launch of class k is bracketed by a pair of
 * hipEvents from now on (mask 0 switches it off; calling it also clears earlier records).  Classes:
 *   0 igemm_kernel<T,128|256,128|256,...> (conv fwd + dgrad, >=128 output channels)   1 igemm_kernel<T,128,64,...> (incl. stem)
 *   2 wgrad_kernel<T,128,*>   3 wgrad_kernel<T,64,*> (incl. stem)   4 bn_reduce_kernel (stats + bwd sums)
 *   5 bn_apply_kernel   6 other (ingest, pools, FC, weight prep)   7 bn_bwd_apply_kernel
 *   profile_read kind 8: the 3x3 convolutions (fwd + dgrad) among the recorded launches of classes 0 and 1
 *   profile_read kind 9: the launches that ran on e4m3 operands (fp8 ctx) among the recorded launches
 * (one class = one kernel symbol, so the averages can be checked against rocprofv3 --kernel-trace --stats)
 * mi355_resnet50_profile_read(ctx, k, ...) waits for the recorded events of class k and returns their
 * summed duration (ms), launch count and the algorithmic FLOPs / bytes those launches represent.    */
int mi355_resnet50_profile(mi355_ctx* ctx, int class_mask);
int mi355_resnet50_profile_read(mi355_ctx* ctx, int kind, double* total_ms, int* launches,
                                double* alg_flops, double* alg_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MI355RN_H */
