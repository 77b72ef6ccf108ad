"""ORACLE (test infrastructure — never imported by the product path).

Per-op CPU restatement of the arithmetic the reference's hot path dispatches to, written on plain
torch-CPU fp32 ops (the "reference's own torch CPU path" of BASELINE.json's north_star).  The reference
itself contains no arithmetic for this path — it calls into un-vendored packages:
  model   : pytorch_tools.models.resnet50   (train.py:64, configs/hydra_exp/1.r50_baseline.yaml:22-23)
  loss    : pytorch_tools.losses.smooth.CrossEntropyLoss (sota_imagenet/arg_parser.py:140-142)
  optim   : torch.optim._multi_tensor.SGD   (sota_imagenet/arg_parser.py:136-138)
and it ships no tests / golden vectors for it  =>  PARITY UNPINNED by the reference:
the oracle is pinned only by torch's own CPU kernels in this container (torch 2.10 CPU, oneDNN/MKL) and by
the fixtures under tests/golden/ generated from them (tests/golden/make_golden.py).

All functions take/return NHWC tensors (the layout of the native path) and convert to torch's NCHW inside.
"""
import torch
import torch.nn.functional as F


def nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def krsc_to_oihw(w):
    return w.permute(0, 3, 1, 2).contiguous()


def oihw_to_krsc(w):
    return w.permute(0, 2, 3, 1).contiguous()


def conv2d_fwd(x_nhwc, w_krsc, stride, pad):
    """K2 of SURVEY §2.3; call form sota_imagenet/callbacks.py:316."""
    y = F.conv2d(nhwc_to_nchw(x_nhwc.float()), krsc_to_oihw(w_krsc.float()), stride=stride, padding=pad)
    return nchw_to_nhwc(y)


def conv2d_bwd(x_nhwc, w_krsc, dy_nhwc, stride, pad):
    """dgrad + wgrad through torch autograd (K8; sota_imagenet/callbacks.py:317). Returns (dx NHWC, dw KRSC)."""
    x = nhwc_to_nchw(x_nhwc.float()).requires_grad_(True)
    w = krsc_to_oihw(w_krsc.float()).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    y.backward(nhwc_to_nchw(dy_nhwc.float()))
    return nchw_to_nhwc(x.grad), oihw_to_krsc(w.grad)


def bn_train(x_nhwc, gamma, beta, running_mean, running_var, residual=None, relu=True, eps=1e-5, momentum=0.1):
    """BatchNorm2d training forward (+residual, +ReLU).  K3/K4; momentum per train.py:76.
    Returns (out NHWC, new_running_mean, new_running_var, save_mean, save_invstd)."""
    x = nhwc_to_nchw(x_nhwc.float())
    rm, rv = running_mean.clone().float(), running_var.clone().float()
    y = F.batch_norm(x, rm, rv, gamma.float(), beta.float(), training=True, momentum=momentum, eps=eps)
    if residual is not None:
        y = y + nhwc_to_nchw(residual.float())
    if relu:
        y = F.relu(y)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    return nchw_to_nhwc(y), rm, rv, mean, 1.0 / torch.sqrt(var + eps)


def bn_eval(x_nhwc, gamma, beta, running_mean, running_var, residual=None, relu=True, eps=1e-5):
    x = nhwc_to_nchw(x_nhwc.float())
    y = F.batch_norm(x, running_mean.float(), running_var.float(), gamma.float(), beta.float(), training=False, eps=eps)
    if residual is not None:
        y = y + nhwc_to_nchw(residual.float())
    if relu:
        y = F.relu(y)
    return nchw_to_nhwc(y)


def bn_train_bwd(x_nhwc, gamma, beta, dout_nhwc, residual=None, relu=True, eps=1e-5):
    """Backward of bn_train through autograd. Returns (dx NHWC, dgamma, dbeta, dresidual NHWC or None)."""
    x = nhwc_to_nchw(x_nhwc.float()).requires_grad_(True)
    g = gamma.float().clone().requires_grad_(True)
    b = beta.float().clone().requires_grad_(True)
    r = None
    y = F.batch_norm(x, None, None, g, b, training=True, eps=eps)
    if residual is not None:
        r = nhwc_to_nchw(residual.float()).requires_grad_(True)
        y = y + r
    if relu:
        y = F.relu(y)
    y.backward(nhwc_to_nchw(dout_nhwc.float()))
    return nchw_to_nhwc(x.grad), g.grad, b.grad, (nchw_to_nhwc(r.grad) if r is not None else None)


def maxpool(x_nhwc):
    """MaxPool2d(3, 2, 1) forward (K5). Returns (y NHWC, flat argmax indices NCHW-plane as torch returns them)."""
    y, idx = F.max_pool2d(nhwc_to_nchw(x_nhwc.float()), 3, 2, 1, return_indices=True)
    return nchw_to_nhwc(y), idx


def maxpool_bwd(x_nhwc, dy_nhwc):
    x = nhwc_to_nchw(x_nhwc.float()).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    y.backward(nhwc_to_nchw(dy_nhwc.float()))
    return nchw_to_nhwc(x.grad)


def gap(x_nhwc):
    return x_nhwc.float().mean(dim=(1, 2))


def gap_bwd(dpooled, shape):
    N, H, W, C = shape
    return (dpooled.float() / (H * W)).view(N, 1, 1, C).expand(N, H, W, C).contiguous()


def smooth_ce(logits, target, smoothing):
    """Label-smoothed CE on float (one-hot / soft) targets, reduction mean — K7.
    Restates pytorch_tools.losses.smooth.CrossEntropyLoss as SURVEY.md Appendix C records it
    (call site sota_imagenet/arg_parser.py:140-142, smoothing 0.1 in 1.r50_baseline.yaml:34-35)."""
    logp = F.log_softmax(logits.float(), dim=1)
    nll = -(logp * target.float()).sum(1)
    uni = -logp.mean(1)
    return ((1.0 - smoothing) * nll + smoothing * uni).mean()


def smooth_ce_bwd(logits, target, smoothing):
    z = logits.float().clone().requires_grad_(True)
    loss = smooth_ce(z, target, smoothing)
    loss.backward()
    return loss.detach(), z.grad


def sgd_steps(p, grads, lr, momentum, weight_decay, steps=None):
    """torch.optim.SGD (dampening 0, no nesterov) applied for len(grads) steps — K10.
    Returns (p, momentum_buffer) after the steps."""
    p = p.float().clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=lr, momentum=momentum, weight_decay=weight_decay)
    for g in grads:
        p.grad = g.float().clone()
        opt.step()
    st = opt.state[p]
    buf = st.get("momentum_buffer", None)
    return p.detach(), (buf.clone() if buf is not None else torch.zeros_like(p))


def accuracy(logits, target_onehot, k):
    """top-k accuracy in percent, target = argmax of the one-hot (K12; train.py:130)."""
    tgt = target_onehot.argmax(1)
    topk = logits.float().topk(k, dim=1).indices
    return (topk == tgt[:, None]).any(1).float().mean() * 100.0


# ---- BResNet-50 variant blocks (BASELINE configs[3]; configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51) --------
# The ops live in un-vendored pytorch_tools (modules/residual.py, modules/pooling.py BlurPool, modules/weight_standartization.py):
# restated from their published definitions as SURVEY.md Appendix C records them — UPSTREAM-RECALLED, parity unpinned.
LEAKY = 0.01


def act(x, code):
    """activation codes of the native path: 0 identity, 1 ReLU, 2 leaky ReLU (slope 0.01)"""
    return x if code == 0 else (F.relu(x) if code == 1 else F.leaky_relu(x, LEAKY))


def bn_act_train(x_nhwc, gamma, beta, running_mean, running_var, residual=None, code=1, eps=1e-5, momentum=0.1):
    """bn_train with an activation code (ABN / InplaceABN with norm_act leaky_relu)"""
    x = nhwc_to_nchw(x_nhwc.float())
    rm, rv = running_mean.clone().float(), running_var.clone().float()
    y = F.batch_norm(x, rm, rv, gamma.float(), beta.float(), training=True, momentum=momentum, eps=eps)
    if residual is not None:
        y = y + nhwc_to_nchw(residual.float())
    return nchw_to_nhwc(act(y, code)), rm, rv


def blurpool(x_nhwc):
    """anti-aliased down-sampling (Zhang 2019, as pytorch_tools BlurPool): reflect pad 1, 3x3 binomial / 16, stride 2, depthwise"""
    x = nhwc_to_nchw(x_nhwc.float())
    C = x.shape[1]
    f = torch.tensor([1.0, 2.0, 1.0])
    k = (f[:, None] * f[None, :] / 16.0)[None, None].repeat(C, 1, 1, 1)
    return nchw_to_nhwc(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), k, stride=2, groups=C))


def avgpool2(x_nhwc):
    return nchw_to_nhwc(F.avg_pool2d(nhwc_to_nchw(x_nhwc.float()), 2, 2))


def maxpool3s1(x_nhwc):
    return nchw_to_nhwc(F.max_pool2d(nhwc_to_nchw(x_nhwc.float()), 3, 1, 1))


def eca(x_nhwc, w):
    """ECA: GAP -> conv1d(k, zero pad, no bias) over the channel axis -> sigmoid -> scale"""
    x = x_nhwc.float()
    N, H, W, C = x.shape
    pooled = x.mean(dim=(1, 2))
    k = w.numel()
    z = F.conv1d(pooled.view(N, 1, C), w.float().view(1, 1, k), padding=k // 2).view(N, C)
    return x * torch.sigmoid(z).view(N, 1, 1, C)


def weight_std(w_krsc, eps=1e-5):
    """per output channel (w - mean) / sqrt(var + eps), biased variance over (KH, KW, Cin)"""
    w = w_krsc.float()
    var, mean = torch.var_mean(w, dim=(1, 2, 3), keepdim=True, unbiased=False)
    return (w - mean) / torch.sqrt(var + eps)


def residual_act(branch, scale_n, shortcut, code):
    b = branch.float()
    if scale_n is not None:
        b = b * scale_n.float().view(-1, 1, 1, 1)
    if shortcut is not None:
        b = b + shortcut.float()
    return act(b, code)


def grads(fn, inputs, dout):
    """autograd helper: gradients of fn(*inputs) w.r.t. every input under the upstream gradient dout"""
    xs = [t.detach().float().clone().requires_grad_(True) for t in inputs]
    y = fn(*xs)
    y.backward(dout.float())
    return [t.grad for t in xs]


# ---- fp8 (OCP e4m3fn) operand form of the convolution (BASELINE.json configs[4]; csrc/fp8.hip) -----------------------------
def quantize_e4m3(x, scale=1.0):
    """saturating round-to-nearest-even of x*scale onto the e4m3fn grid (max 448, min subnormal 2^-9), returned as the fp32
    VALUES on that grid.  Restated from the format definition (OCP 8-bit floating point spec v1.0 §5: bias 7, 3 mantissa
    bits, no infinities); tests/test_fp8_oracle.py pins it to torch's own float8_e4m3fn cast."""
    v = (x.float() * scale).double().clamp(-448.0, 448.0)  # the product is taken in fp32, then rounded ONCE onto the grid
    a = v.abs()
    e = torch.floor(torch.log2(a.clamp_min(2.0 ** -20))).clamp_min(-6.0)  # subnormals share the 2^-6 binade
    step = torch.pow(2.0, e - 3.0)
    q = torch.round(a / step) * step  # torch.round is half-to-even
    return (torch.sign(v) * q).float()


def e4m3_bits(vals):
    """the byte encoding of values already on the e4m3fn grid"""
    return vals.to(torch.float8_e4m3fn).view(torch.uint8)


def conv2d_fwd_fp8(xq_vals, wq_vals, stride, pad, oscale):
    """what mi355_conv2d_fwd_fp8 computes: products of grid values (exact in fp32), fp32 accumulation, * oscale"""
    return conv2d_fwd(xq_vals, wq_vals, stride, pad) * oscale


def conv2d_dgrad_fp8(dyq_vals, wq_vals, x_shape, stride, pad, oscale):
    dx, _ = conv2d_bwd(torch.zeros(x_shape), wq_vals, dyq_vals, stride, pad)
    return dx * oscale
