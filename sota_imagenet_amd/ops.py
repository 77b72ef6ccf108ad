"""Per-op Python entry points over the C-ABI (include/mi355rn.h) on PyTorch-ROCm tensors.

These are the unit-level handles the parity tests use; the training path goes through the whole-network
executor (sota_imagenet_amd/models.py).  Tensors are NHWC (`[N,H,W,C]`), conv weights KRSC (`[Cout,KH,KW,Cin]`).
Every function enqueues on torch's current stream and raises RuntimeError on any native failure.
"""
import ctypes

import torch

from . import native
from .native import check, cur_stream, dtype_code, ptr


def _L():
    return native.lib()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise ValueError("tensors must be contiguous CUDA (ROCm) tensors")


def _out_dim(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def conv2d_fwd(x, w, stride=1, pad=0, stats=False):
    """stats=True: returns (y, partial) — `partial` = the BN batch-statistics rows the conv epilogue summed over y, an fp32
    tensor [nblk, 2, Cout] (or None where the launch could not carry them) to hand to bn_fwd_train(y, ..., stats=partial):
    an explicit value owned by the caller, valid for exactly this y as long as y is not modified."""
    _need_cuda(x, w)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    y = torch.empty((N, _out_dim(H, KH, stride, pad), _out_dim(W, KW, stride, pad), Cout), dtype=x.dtype, device=x.device)
    if not stats:
        check(_L().mi355_conv2d_fwd(dtype_code(x.dtype), ptr(x), ptr(w), ptr(y), N, H, W, Cin, Cout, KH, KW, stride, pad, cur_stream()))
        return y
    # rows of the statistics scratch: the implicit-GEMM kernels write <= 768, the generated kernels one per pixel tile (3584 for a
    # 56 x 56 x 64 conv at batch 256, 14336 for a 112 x 112 x 64 one: two-row tiles); the library takes the kernel the buffer allows (the static
    # executors give it their widest layer's)
    rows = 16384 if Cout <= 128 else (8192 if Cout <= 256 else 768)
    partial = torch.empty(rows * 2 * Cout, dtype=torch.float32, device=x.device)
    nblk = ctypes.c_int(0)
    check(_L().mi355_conv2d_fwd_stats(dtype_code(x.dtype), ptr(x), ptr(w), ptr(y), ptr(partial), partial.numel() * 4, ctypes.byref(nblk), N, H, W, Cin,
                                      Cout, KH, KW, stride, pad, cur_stream()))
    return y, (partial[: nblk.value * 2 * Cout].view(nblk.value, 2, Cout) if nblk.value > 0 else None)


def conv2d_fwd_bn_in(y_in, scale, shift, w, pad=1):
    """the forward conv with the previous layer's BatchNorm + ReLU in its operand path (mi355_conv2d_fwd_bn_in): returns
    (y, partial, a, a_bits) — y = conv(a, w) with a = relu(y_in * scale + shift) rounded to y_in's dtype, the BN statistics rows of y, and
    the by-products a and its ReLU bits (1 byte per 8 channels) the same launch leaves for the backward pass"""
    _need_cuda(y_in, scale, shift, w)
    N, H, W, Cin = y_in.shape
    Cout, KH, KW, _ = w.shape
    y = torch.empty((N, H, W, Cout), dtype=y_in.dtype, device=y_in.device)
    a = torch.empty_like(y_in)
    bits = torch.empty((N, H, W, Cin // 8), dtype=torch.uint8, device=y_in.device)
    ss = torch.cat([scale.float().reshape(-1), shift.float().reshape(-1)]).contiguous()
    rows = 8192 if Cout <= 256 else 768
    partial = torch.empty(rows * 2 * Cout, dtype=torch.float32, device=y_in.device)
    nblk = ctypes.c_int(0)
    check(_L().mi355_conv2d_fwd_bn_in(dtype_code(y_in.dtype), ptr(y_in), ptr(ss), ptr(w), ptr(y), ptr(a), ptr(bits), ptr(partial), partial.numel() * 4,
                                      ctypes.byref(nblk), N, H, W, Cin, Cout, KH, KW, 1, pad, cur_stream()))
    return y, (partial[: nblk.value * 2 * Cout].view(nblk.value, 2, Cout) if nblk.value > 0 else None), a, bits


def quantize_fp8(x, scale=1.0):
    """q = saturate_e4m3fn(x * scale) as a torch.float8_e4m3fn tensor of x's shape (x fp32 or bf16, numel % 8 == 0)."""
    _need_cuda(x)
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_L().mi355_quantize_fp8(dtype_code(x.dtype), ptr(x), ptr(q), float(scale), x.numel(), cur_stream()))
    return q.view(torch.float8_e4m3fn)


def conv2d_fwd_fp8(xq, wq, stride=1, pad=0, oscale=1.0):
    """bf16 y = conv(xq NHWC e4m3, wq KRSC e4m3) * oscale on the fp8 MFMA path (Cin, Cout multiples of 128)."""
    _need_cuda(xq, wq)
    N, H, W, Cin = xq.shape
    Cout, KH, KW, _ = wq.shape
    y = torch.empty((N, _out_dim(H, KH, stride, pad), _out_dim(W, KW, stride, pad), Cout), dtype=torch.bfloat16, device=xq.device)
    check(_L().mi355_conv2d_fwd_fp8(ptr(xq), ptr(wq), ptr(y), float(oscale), N, H, W, Cin, Cout, KH, KW, stride, pad, cur_stream()))
    return y


def conv2d_dgrad_fp8(dyq, wq, x_shape, stride=1, pad=0, oscale=1.0, wt=None):
    """bf16 dx for the conv above from e4m3 dy and the KRSC e4m3 weights (transposed here to [Cin][KH][KW][Cout])."""
    _need_cuda(dyq, wq)
    N, H, W, Cin = x_shape
    Cout, KH, KW, _ = wq.shape
    if wt is None:
        wt = wq.view(torch.uint8).permute(3, 1, 2, 0).contiguous()
    dx = torch.empty((N, H, W, Cin), dtype=torch.bfloat16, device=dyq.device)
    check(_L().mi355_conv2d_dgrad_fp8(ptr(dyq), ptr(wt), ptr(dx), float(oscale), N, H, W, Cin, Cout, KH, KW, stride, pad, cur_stream()))
    return dx


_WGRAD_WS = {}


def conv2d_wgrad_fp8(dyq, xq, KH, KW, stride=1, pad=0, oscale=1.0):
    """fp32 dw [Cout,KH,KW,Cin] = oscale * sum_pixels dyq (x) xq from e4m3 dy (NHWC) and x (NHWC) on the fp8 MFMA path."""
    _need_cuda(dyq, xq)
    N, H, W, Cin = xq.shape
    Cout = dyq.shape[-1]
    n = _L().mi355_conv2d_workspace_bytes(native.BF16, N, H, W, Cin, Cout, KH, KW, stride, pad)
    # split-K slabs + the two scale words: one workspace per (device, stream, size) — calls on different streams must not share slabs —
    # and at most 8 of them (the oldest goes first)
    key = (dyq.device, torch.cuda.current_stream(dyq.device).cuda_stream, n)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        while len(_WGRAD_WS) >= 8:
            _WGRAD_WS.pop(next(iter(_WGRAD_WS)))
        ws = _WGRAD_WS[key] = torch.empty(n + 256, dtype=torch.uint8, device=dyq.device)
    dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.float32, device=dyq.device)
    check(_L().mi355_conv2d_wgrad_fp8(ptr(dyq), ptr(xq), ptr(dw), 0.0, float(oscale), N, H, W, Cin, Cout, KH, KW, stride, pad, ptr(ws), n + 256, cur_stream()))
    return dw


def ingest_u8(packed, table_host, table_dev, S, mean=127.5, std=51.0, aug_host=None, aug_dev=None):
    """decoded u8 RGB crops (packed device bytes + mi355_crop table as a numpy structured array and its device copy) ->
    fp32 NCHW [N,3,S,S]: resize, window, [augment table: blur / colour / grey / erase], mirror, normalise (csrc/ingest.hip)."""
    _need_cuda(packed, table_dev, aug_dev)
    N = int(table_host.shape[0])
    if table_host.dtype.itemsize != 40 or table_dev.numel() * table_dev.element_size() != 40 * N or not table_host.flags["C_CONTIGUOUS"]:
        raise ValueError("ingest_u8: the descriptor table must be N contiguous 40-byte mi355_crop records on both sides")
    out = torch.empty((N, 3, S, S), dtype=torch.float32, device=packed.device)
    if aug_host is None:
        check(_L().mi355_ingest_u8(ptr(packed), packed.numel(), table_host.ctypes.data, ptr(table_dev), N, int(S), float(mean), float(std), ptr(out), cur_stream()))
        return out
    if aug_host.dtype.itemsize != 128 or aug_host.shape[0] != N or aug_dev is None or aug_dev.numel() * aug_dev.element_size() != 128 * N or not aug_host.flags["C_CONTIGUOUS"]:
        raise ValueError("ingest_u8: the augment table must be N contiguous 128-byte mi355_augment records on both sides")
    scratch = torch.empty((N, 3, S, S), dtype=torch.float32, device=packed.device) if bool((aug_host["blur_sigma"] > 0).any()) else None
    check(_L().mi355_ingest_u8_aug(ptr(packed), packed.numel(), table_host.ctypes.data, ptr(table_dev), aug_host.ctypes.data, ptr(aug_dev), N, int(S),
                                   float(mean), float(std), ptr(scratch), ptr(out), cur_stream()))
    return out


def _conv_ws(dt, N, H, W, Cin, Cout, KH, KW, stride, pad, device):
    n = _L().mi355_conv2d_workspace_bytes(dt, N, H, W, Cin, Cout, KH, KW, stride, pad)
    return torch.empty(n, dtype=torch.uint8, device=device), n


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0, addend=None):
    _need_cuda(dy, w, addend)
    N, H, W, Cin = x_shape
    Cout, KH, KW, _ = w.shape
    dt = dtype_code(dy.dtype)
    ws, n = _conv_ws(dt, N, H, W, Cin, Cout, KH, KW, stride, pad, dy.device)
    dx = torch.empty((N, H, W, Cin), dtype=dy.dtype, device=dy.device)
    check(_L().mi355_conv2d_dgrad(dt, ptr(dy), ptr(w), ptr(dx), ptr(addend), N, H, W, Cin, Cout, KH, KW, stride, pad, ptr(ws), n, cur_stream()))
    return dx


def conv2d_dgrad_bn(dy, w, x_shape, stride=1, pad=0, addend=None, addend_bits=None, bn_y=None, bn_bits=None, bn_mean=None, bn_invstd=None, addend_sub2=False, slope=0.0):
    """the data gradient as the executor's backward launches it: dx = dgrad(dy) + addend (under the ReLU bits addend_bits), and — with
    bn_y / bn_bits / bn_mean / bn_invstd — the BN-backward partial rows [nblk][2][Cin] (sum dz, sum dz * xhat) of the layer dx is the
    activation gradient of.  addend_sub2: the addend is [N, H/2, W/2, Cin], standing for a full-resolution tensor that is zero at odd rows /
    columns.  slope: the sums under a leaky-ReLU mask (dz = dx * slope where the bit is clear; 0.01 at BResNet-50's shapes, mi355_conv2d_dgrad_bn_leaky).
    Returns (dx, partial or None)."""
    _need_cuda(dy, w, addend, addend_bits, bn_y, bn_bits, bn_mean, bn_invstd)
    N, H, W, Cin = x_shape
    Cout, KH, KW, _ = w.shape
    dt = dtype_code(dy.dtype)
    ws, n = _conv_ws(dt, N, H, W, Cin, Cout, KH, KW, stride, pad, dy.device)
    dx = torch.empty((N, H, W, Cin), dtype=dy.dtype, device=dy.device)
    part = None
    nblk = ctypes.c_int(0)
    if bn_y is not None:
        part = torch.empty((4096, 2, Cin), dtype=torch.float32, device=dy.device)
    check(_L().mi355_conv2d_dgrad_bn_leaky(dt, ptr(dy), ptr(w), ptr(dx), ptr(addend), ptr(addend_bits), int(addend_sub2), ptr(bn_y), ptr(bn_bits), ptr(bn_mean),
                                           ptr(bn_invstd), float(slope), ptr(part), 0 if part is None else part.numel() * 4, ctypes.byref(nblk), N, H, W, Cin, Cout,
                                           KH, KW, stride, pad, ptr(ws), n, cur_stream()))
    if part is not None:
        part = part[:nblk.value] if nblk.value > 0 else None
    return dx, part


def last_conv_kernel():
    """name of the kernel the last conv / weight-gradient launch of this thread went to (mi355_last_conv_kernel)"""
    return _L().mi355_last_conv_kernel().decode()


def conv2d_wgrad(dy, x, KH, KW, stride=1, pad=0, dw=None, beta=0.0):
    _need_cuda(dy, x, dw)
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    dt = dtype_code(dy.dtype)
    ws, n = _conv_ws(dt, N, H, W, Cin, Cout, KH, KW, stride, pad, dy.device)
    if dw is None:
        dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.float32, device=dy.device)
    check(_L().mi355_conv2d_wgrad(dt, ptr(dy), ptr(x), ptr(dw), beta, N, H, W, Cin, Cout, KH, KW, stride, pad, ptr(ws), n, cur_stream()))
    return dw


def stem_ingest(x_nchw, dtype):
    _need_cuda(x_nchw)
    N, _, H, W = x_nchw.shape
    dt = dtype_code(dtype)
    nbytes = _L().mi355_stem_xpad_bytes(dt, N, H, W)
    xpad = torch.zeros(nbytes, dtype=torch.uint8, device=x_nchw.device)
    check(_L().mi355_stem_ingest(dt, ptr(x_nchw), ptr(xpad), N, H, W, cur_stream()))
    return xpad


def stem_fwd(xpad, w, N, H, W, dtype):
    """w: [64,7,7,3] fp32 (KRSC)."""
    _need_cuda(xpad, w)
    dt = dtype_code(dtype)
    n = _L().mi355_stem_workspace_bytes(dt, N, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device=w.device)
    y = torch.empty((N, H // 2, W // 2, 64), dtype=dtype, device=w.device)
    check(_L().mi355_stem_fwd(dt, ptr(xpad), ptr(w), ptr(y), N, H, W, ptr(ws), n, cur_stream()))
    return y


def stem_wgrad(dy, xpad, N, H, W):
    _need_cuda(dy, xpad)
    dt = dtype_code(dy.dtype)
    n = _L().mi355_stem_workspace_bytes(dt, N, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device=dy.device)
    dw = torch.empty((64, 7, 7, 3), dtype=torch.float32, device=dy.device)
    check(_L().mi355_stem_wgrad(dt, ptr(dy), ptr(xpad), ptr(dw), 0.0, N, H, W, ptr(ws), n, cur_stream()))
    return dw


def _bn_ws(C, device):
    n = _L().mi355_bn_workspace_bytes(C)
    return torch.empty(n, dtype=torch.uint8, device=device), n


def bn_fwd_train(x, gamma, beta, running_mean, running_var, residual=None, relu=True, eps=1e-5, momentum=0.1, stats=None):
    """x: [..., C] NHWC.  Updates running stats in place.  Returns (out, save_mean, save_invstd).
    stats: the partial rows conv2d_fwd(stats=True) returned for THIS x (no reduction pass over x then)."""
    _need_cuda(x, gamma, beta, running_mean, running_var, residual)
    C = x.shape[-1]
    M = x.numel() // C
    out = torch.empty_like(x)
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    si = torch.empty(C, dtype=torch.float32, device=x.device)
    if stats is not None:  # the conv that produced x already summed it
        _need_cuda(stats)
        if stats.dim() != 3 or stats.shape[1:] != (2, C) or stats.dtype != torch.float32:
            raise ValueError(f"bn_fwd_train: stats must be the fp32 [nblk, 2, {C}] rows conv2d_fwd(stats=True) returned for this tensor")
        ws = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        check(_L().mi355_bn_fwd_train_partial(dtype_code(x.dtype), ptr(x), ptr(residual), ptr(out), ptr(gamma), ptr(beta), ptr(running_mean),
                                              ptr(running_var), ptr(sm), ptr(si), M, C, eps, momentum, int(relu), ptr(stats), stats.shape[0], ptr(ws),
                                              ws.numel() * 4, cur_stream()))
        return out, sm, si
    ws, n = _bn_ws(C, x.device)
    check(_L().mi355_bn_fwd_train(dtype_code(x.dtype), ptr(x), ptr(residual), ptr(out), ptr(gamma), ptr(beta), ptr(running_mean),
                                  ptr(running_var), ptr(sm), ptr(si), M, C, eps, momentum, int(relu), ptr(ws), n, cur_stream()))
    return out, sm, si


def bn_fwd_eval(x, gamma, beta, running_mean, running_var, residual=None, relu=True, eps=1e-5):
    _need_cuda(x, gamma, beta, running_mean, running_var, residual)
    C = x.shape[-1]
    M = x.numel() // C
    out = torch.empty_like(x)
    ws, n = _bn_ws(C, x.device)
    check(_L().mi355_bn_fwd_eval(dtype_code(x.dtype), ptr(x), ptr(residual), ptr(out), ptr(gamma), ptr(beta), ptr(running_mean),
                                 ptr(running_var), M, C, eps, int(relu), ptr(ws), n, cur_stream()))
    return out


def bn_bwd(dout, out, x, gamma, save_mean, save_invstd, relu=True, want_dz=False):
    """Returns (dx, dgamma, dbeta, dz or None)."""
    _need_cuda(dout, out, x, gamma, save_mean, save_invstd)
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    dz = torch.empty_like(x) if want_dz else None
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    ws, n = _bn_ws(C, x.device)
    check(_L().mi355_bn_bwd(dtype_code(x.dtype), ptr(dout), ptr(out), ptr(x), ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(dx),
                            ptr(dz), ptr(dg), ptr(db), 0.0, M, C, int(relu), ptr(ws), n, cur_stream()))
    return dx, dg, db, dz


def maxpool_fwd(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((N, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)
    check(_L().mi355_maxpool_fwd(dtype_code(x.dtype), ptr(x), ptr(y), ptr(idx), N, H, W, C, cur_stream()))
    return y, idx


def maxpool_bwd(dy, idx, x_shape):
    _need_cuda(dy, idx)
    N, H, W, C = x_shape
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    check(_L().mi355_maxpool_bwd(dtype_code(dy.dtype), ptr(dy), ptr(idx), ptr(dx), N, H, W, C, cur_stream()))
    return dx


def gap_fwd(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    pooled = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(_L().mi355_gap_fwd(dtype_code(x.dtype), ptr(x), ptr(pooled), N, H * W, C, cur_stream()))
    return pooled


def gap_bwd(dpooled, x_shape, dtype):
    _need_cuda(dpooled)
    N, H, W, C = x_shape
    dx = torch.empty((N, H, W, C), dtype=dtype, device=dpooled.device)
    check(_L().mi355_gap_bwd(dtype_code(dtype), ptr(dpooled), ptr(dx), N, H * W, C, cur_stream()))
    return dx


def ce_loss(logits, target, smoothing=0.0, grad_scale=1.0, need_grad=True):
    """Returns (loss scalar tensor on device, dlogits or None)."""
    _need_cuda(logits, target)
    N, C = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    row = torch.empty(N, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if need_grad else None
    check(_L().mi355_ce_loss(ptr(logits), ptr(target), smoothing, grad_scale, ptr(loss), ptr(row), ptr(dl), N, C, cur_stream()))
    return loss, dl


def sgd_step(p, g, m, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0, ema=None, ema_decay=0.0):
    """ema (optional, same shape as p): the moving average of the updated parameters, advanced in the same kernel (mi355_sgd_step_ema)"""
    _need_cuda(p, g, m)
    if ema is None:
        check(_L().mi355_sgd_step(ptr(p), ptr(g), ptr(m), p.numel(), lr, momentum, weight_decay, grad_scale, cur_stream()))
        return
    _need_cuda(ema)
    assert ema.numel() == p.numel() and ema.dtype == torch.float32 and ema.is_contiguous()
    check(_L().mi355_sgd_step_ema(ptr(p), ptr(g), ptr(m), ptr(ema), p.numel(), lr, momentum, weight_decay, grad_scale, ema_decay, cur_stream()))


# ---- BResNet-50 variant blocks (include/mi355rn.h, csrc/variant.hip) ---------------------------------------------------
def blurpool_fwd(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    check(_L().mi355_blurpool_fwd(dtype_code(x.dtype), ptr(x), ptr(y), N, H, W, C, cur_stream()))
    return y


def blurpool_bwd(dy, x_shape):
    _need_cuda(dy)
    N, H, W, C = x_shape
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    check(_L().mi355_blurpool_bwd(dtype_code(dy.dtype), ptr(dy), ptr(dx), N, H, W, C, cur_stream()))
    return dx


def avgpool2_fwd(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    check(_L().mi355_avgpool2_fwd(dtype_code(x.dtype), ptr(x), ptr(y), N, H, W, C, cur_stream()))
    return y


def avgpool2_bwd(dy, x_shape):
    _need_cuda(dy)
    N, H, W, C = x_shape
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    check(_L().mi355_avgpool2_bwd(dtype_code(dy.dtype), ptr(dy), ptr(dx), N, H, W, C, cur_stream()))
    return dx


def maxpool3s1_fwd(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    idx = torch.empty((N, H, W, C), dtype=torch.uint8, device=x.device)
    check(_L().mi355_maxpool3s1_fwd(dtype_code(x.dtype), ptr(x), ptr(y), ptr(idx), N, H, W, C, cur_stream()))
    return y, idx


def maxpool3s1_bwd(dy, idx):
    _need_cuda(dy, idx)
    N, H, W, C = dy.shape
    dx = torch.empty_like(dy)
    check(_L().mi355_maxpool3s1_bwd(dtype_code(dy.dtype), ptr(dy), ptr(idx), ptr(dx), N, H, W, C, cur_stream()))
    return dx


def eca_fwd(x, w):
    """returns (y, pooled [N,C] fp32, gate [N,C] fp32)"""
    _need_cuda(x, w)
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    pooled = torch.empty((N, C), dtype=torch.float32, device=x.device)
    gate = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(_L().mi355_eca_fwd(dtype_code(x.dtype), ptr(x), ptr(w), w.numel(), ptr(y), ptr(pooled), ptr(gate), N, H * W, C, cur_stream()))
    return y, pooled, gate


def eca_bwd(dy, x, w, pooled, gate):
    """returns (dx, dw [k] fp32)"""
    _need_cuda(dy, x, w, pooled, gate)
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    ws = torch.empty(2 * N * C + 128 * 9, dtype=torch.float32, device=x.device)  # mi355_eca_bwd: sprod, dpool, dw partials
    check(_L().mi355_eca_bwd(dtype_code(x.dtype), ptr(dy), ptr(x), ptr(w), w.numel(), ptr(pooled), ptr(gate), ptr(dx), ptr(dw), 0.0, ptr(ws),
                             N, H * W, C, cur_stream()))
    return dx, dw


def weight_std_fwd(w, eps=1e-5):
    """w: [Cout, ...] fp32 contiguous.  returns (w_hat, invstd [Cout])"""
    _need_cuda(w)
    Cout = w.shape[0]
    K = w.numel() // Cout
    w_hat = torch.empty_like(w)
    mean = torch.empty(Cout, dtype=torch.float32, device=w.device)
    invstd = torch.empty(Cout, dtype=torch.float32, device=w.device)
    check(_L().mi355_weight_std_fwd(ptr(w), ptr(w_hat), ptr(mean), ptr(invstd), Cout, K, eps, cur_stream()))
    return w_hat, invstd


def weight_std_bwd(dw_hat, w_hat, invstd):
    _need_cuda(dw_hat, w_hat, invstd)
    Cout = w_hat.shape[0]
    dw = torch.empty_like(w_hat)
    check(_L().mi355_weight_std_bwd(ptr(dw_hat), ptr(w_hat), ptr(invstd), ptr(dw), 0.0, Cout, w_hat.numel() // Cout, cur_stream()))
    return dw


def residual_act_fwd(branch, shortcut=None, scale_n=None, act=1):
    _need_cuda(branch, shortcut, scale_n)
    N = branch.shape[0]
    out = torch.empty_like(branch)
    check(_L().mi355_residual_act_fwd(dtype_code(branch.dtype), ptr(branch), ptr(scale_n), ptr(shortcut), ptr(out), N, branch.numel() // N, act, cur_stream()))
    return out


def residual_act_bwd(dout, out, scale_n=None, act=1, want_shortcut=True):
    _need_cuda(dout, out, scale_n)
    N = out.shape[0]
    db = torch.empty_like(out)
    ds = torch.empty_like(out) if want_shortcut else None
    check(_L().mi355_residual_act_bwd(dtype_code(out.dtype), ptr(dout), ptr(out), ptr(scale_n), ptr(db), ptr(ds), N, out.numel() // N, act, cur_stream()))
    return db, ds


def keep_scale(n, p, seed, counter, device):
    keep = torch.empty(n, dtype=torch.float32, device=device)
    check(_L().mi355_keep_scale(ptr(keep), n, float(p), int(seed), int(counter), cur_stream()))
    return keep
