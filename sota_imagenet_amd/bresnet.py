"""BResNet-50 — the model of BASELINE.json configs[3] — on the native per-op C-ABI (HIP kernels for every tensor op).

The reference builds it with `_target_: pytorch_tools.models.resnet50` and the model_params of
configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51 (stem_type deep, antialias, attn_type eca, norm_layer
inplaceabn, norm_act leaky_relu, drop_rate 0.2, drop_connect_rate 0.2) and wraps every conv in weight standardisation
(train.py:66-67; `weight_standardization: True`, yaml:59).  pytorch_tools is not vendored: the block structure is restated
from its published source as SURVEY.md Appendix C recalls it (the same statement as the oracle, oracle/bresnet50_ref.py):
  deep stem   conv3x3(3,32,s2)+ABN, conv3x3(32,32)+ABN, conv3x3(32,64), bn1 = ABN(64); pool = maxpool 3x3/1 + BlurPool
  bottleneck  conv1x1+ABN, conv3x3 (stride 1)+ABN [+BlurPool when the block strides], conv1x1+ABN(identity), ECA(k=3),
              drop-connect(rate * i / 16) on the branch, + shortcut ([AvgPool 2x2] conv1x1 ABN(identity) where the block
              strides / widens), leaky ReLU
  head        GAP, dropout(drop_rate), FC
Unlike models.ResNet50 (one static C++ executor call per step) this graph is driven from Python, one C-ABI call per op,
through torch.autograd.Function nodes: conv fwd/dgrad/wgrad (conv_igemm*.hip, conv_wgrad.hip), BN+act (bn.hip), and the
variant kernels of csrc/variant.hip.  Activations are NHWC in the compute dtype (bf16 | fp32); parameters are fp32 nn.Parameters
with the names / shapes pytorch_tools gives them (conv1.0.weight, layer1.0.se_module.conv.weight, layer2.0.downsample.0.weight …).
Channel counts below 64 (the 3 / 32-channel stem) run zero-padded to 64, the granule of the conv kernels.  There is no CPU
path: a CPU tensor raises.
"""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from .models import _BNLeaf, _FlatModel, _ResNetFn

LEAKY_ACT = 2  # activation code of the C-ABI: 0 identity, 1 ReLU, 2 leaky ReLU (0.01)


def _pad_c(t, c, dim=-1):
    """zero-pads dimension `dim` of t to c entries"""
    if t.shape[dim] == c:
        return t
    shape = list(t.shape)
    shape[dim] = c - t.shape[dim]
    return torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim=dim)


def _up64(c):
    return (c + 63) // 64 * 64


class _ConvFn(torch.autograd.Function):
    """y = conv(x, WS(w)): weight standardisation (optional), channel padding, cast, conv — one node, weight gradient in fp32"""

    @staticmethod
    def forward(ctx, x, w, stride, standardize, want_stats):
        cout, cin, k, _ = w.shape
        wk = w.detach().permute(0, 2, 3, 1).contiguous()  # KRSC fp32
        if standardize:
            w_hat, invstd = ops.weight_std_fwd(wk)
        else:
            w_hat, invstd = wk, None
        wp = _pad_c(_pad_c(w_hat, _up64(cin), 3), _up64(cout), 0).to(x.dtype).contiguous()
        # want_stats (training): the epilogue also sums y for the BN that follows — handed on as an explicit second output
        y, st = ops.conv2d_fwd(x, wp, stride, k // 2, stats=True) if want_stats else (ops.conv2d_fwd(x, wp, stride, k // 2), None)
        ctx.save_for_backward(x, wp, w_hat, invstd if invstd is not None else torch.empty(0))
        ctx.meta = (cout, cin, k, stride, standardize)
        if st is None:
            st = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(st)
        return y, st

    @staticmethod
    def backward(ctx, dy, _dst):
        x, wp, w_hat, invstd = ctx.saved_tensors
        cout, cin, k, stride, standardize = ctx.meta
        dy = dy.contiguous()
        dx = ops.conv2d_dgrad(dy, wp, tuple(x.shape), stride, k // 2) if ctx.needs_input_grad[0] else None
        dwp = ops.conv2d_wgrad(dy, x, k, k, stride, k // 2)  # fp32 [Cout_p, k, k, Cin_p]
        dw_hat = dwp[:cout, :, :, :cin].contiguous()
        dwk = ops.weight_std_bwd(dw_hat, w_hat, invstd) if standardize else dw_hat
        return dx, dwk.permute(0, 3, 1, 2), None, None, None


class _BNActFn(torch.autograd.Function):
    """ABN: BatchNorm (batch statistics, running stats updated in place) + activation code"""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, act, momentum, training, stats):
        C, Cp = gamma.numel(), x.shape[-1]
        g, b = _pad_c(gamma.detach(), Cp).contiguous(), _pad_c(beta.detach(), Cp).contiguous()
        rmp = _pad_c(rm, Cp).clone()
        rvp = (rv if Cp == C else torch.cat([rv, torch.ones(Cp - C, device=rv.device)])).clone()  # padded channels: var 1
        if not training:
            return ops.bn_fwd_eval(x, g, b, rmp, rvp, relu=act)
        out, mean, invstd = ops.bn_fwd_train(x, g, b, rmp, rvp, relu=act, momentum=momentum, stats=stats if stats is not None and stats.numel() else None)
        rm.copy_(rmp[:C])
        rv.copy_(rvp[:C])
        ctx.save_for_backward(x, out, g, mean, invstd)
        ctx.meta = (C, act)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, g, mean, invstd = ctx.saved_tensors
        C, act = ctx.meta
        dx, dg, db, _ = ops.bn_bwd(dout.contiguous(), out, x, g, mean, invstd, relu=act)
        return dx, dg[:C], db[:C], None, None, None, None, None, None


class _BlurPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.blurpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.blurpool_bwd(dy.contiguous(), ctx.shape)


class _AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.avgpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool2_bwd(dy.contiguous(), ctx.shape)


class _MaxPool3s1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = ops.maxpool3s1_fwd(x)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3s1_bwd(dy.contiguous(), idx)


class _ECAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        wk = w.detach().reshape(-1).contiguous()
        y, pooled, gate = ops.eca_fwd(x, wk)
        ctx.save_for_backward(x, wk, pooled, gate)
        ctx.wshape = tuple(w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wk, pooled, gate = ctx.saved_tensors
        dx, dw = ops.eca_bwd(dy.contiguous(), x, wk, pooled, gate)
        return dx, dw.view(ctx.wshape)


class _ResidualActFn(torch.autograd.Function):
    """out = act(branch * keep[n] + shortcut)"""

    @staticmethod
    def forward(ctx, branch, shortcut, keep, act):
        out = ops.residual_act_fwd(branch, shortcut, keep, act)
        ctx.save_for_backward(out, keep if keep is not None else torch.empty(0))
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dout):
        out, keep = ctx.saved_tensors
        db, ds = ops.residual_act_bwd(dout.contiguous(), out, keep if keep.numel() else None, ctx.act)
        return db, ds, None, None


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        return ops.gap_fwd(x)

    @staticmethod
    def backward(ctx, dp):
        return ops.gap_bwd(dp.contiguous(), ctx.shape, ctx.dtype)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, stride, standardize):
        super().__init__()
        # OIHW shape in channels_last memory = KRSC, the kernels' weight layout (and what the gradient comes back in)
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last))
        self.stride, self.standardize = stride, standardize

    def forward(self, x, stats=False):
        """stats=True (the caller feeds a BatchNorm next, in training): returns (y, partial statistics rows of y)"""
        y, st = _ConvFn.apply(x, self.weight, self.stride, self.standardize, bool(stats))
        return (y, st) if stats else y


class _ABN(nn.Module):
    """BN-like leaf for patch_bn_mom (train.py:76): has `.momentum`, weight / bias / running_mean / running_var / num_batches_tracked"""

    def __init__(self, c, act):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
        self.momentum, self.eps, self.act = 0.1, 1e-5, act

    def forward(self, x, stats=None):
        if self.training:
            self.num_batches_tracked += 1
        return _BNActFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.act, self.momentum, self.training, stats)


def _conv_bn(conv, bn, x):
    """conv -> ABN; in training the conv's epilogue statistics go to the BatchNorm as an explicit value"""
    if bn.training:
        y, st = conv(x, stats=True)
        return bn(y, st)
    return bn(conv(x))


class _ECA(nn.Module):
    def __init__(self, k=3):
        super().__init__()
        self.conv = nn.Module()
        self.conv.weight = nn.Parameter(torch.empty(1, 1, k))

    def forward(self, x):
        return _ECAFn.apply(x, self.conv.weight)


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, has_ds, standardize):
        super().__init__()
        self.conv1, self.bn1 = _Conv(cin, planes, 1, 1, standardize), _ABN(planes, LEAKY_ACT)
        self.conv2, self.bn2 = _Conv(planes, planes, 3, 1, standardize), _ABN(planes, LEAKY_ACT)
        self.conv3, self.bn3 = _Conv(planes, planes * 4, 1, 1, standardize), _ABN(planes * 4, 0)
        self.se_module = _ECA(3)
        self.stride = stride
        self.downsample = None
        if has_ds:
            self.downsample = nn.Sequential()
            self.downsample.add_module("0", _Conv(cin, planes * 4, 1, 1, standardize))
            self.downsample.add_module("1", _ABN(planes * 4, 0))

    def forward(self, x, keep):
        out = _conv_bn(self.conv1, self.bn1, x)
        out = _conv_bn(self.conv2, self.bn2, out)
        if self.stride == 2:
            out = _BlurPoolFn.apply(out)
        out = self.se_module(_conv_bn(self.conv3, self.bn3, out))
        sc = x
        if self.downsample is not None:
            sc = _AvgPool2Fn.apply(x) if self.stride == 2 else x
            sc = _conv_bn(self.downsample[0], self.downsample[1], sc)
        return _ResidualActFn.apply(out, sc, keep, LEAKY_ACT)


class _DropStream:
    """position of the drop-connect / dropout generator, shared by the two BResNet-50 forms: the stream is keyed by (seed, step).
    `reseed` gives every (run seed, rank) its own stream (the reference draws from per-process torch generators, so ranks must not
    share keep vectors); `set_drop_position` makes the counter a function of (epoch, step) so a resumed run continues the sequence
    instead of replaying it.  The Runner calls both (fit_wrapper.Runner.fit / _run_loader); a `seed=` given to the constructor stays."""

    def reseed(self, random_seed, rank):
        if not self._seed_given:
            self.seed = (int(random_seed or 0) * 1000003 + int(rank) * 7919 + 54321) & 0x7FFFFFFF

    def set_drop_position(self, epoch, step=0):
        self._step = (int(epoch) << 32) | int(step)


class BResNet50Graph(_DropStream, nn.Module):
    """the round-2 form: this graph driven from Python, one C-ABI call per op through torch.autograd nodes.  Kept as the
    cross-check of the static executor (tests/test_variant_gpu.py: same kernels in the same order => same bits up to the FC)."""

    def __init__(self, num_classes=1000, dtype="bf16", drop_rate=0.0, drop_connect_rate=0.0, weight_standardization=False, seed=None, **kw):
        super().__init__()
        unknown = set(kw) - {"pretrained", "stem_type", "antialias", "attn_type", "norm_layer", "norm_act"}
        if unknown:
            raise TypeError(f"bresnet50: unsupported arguments {sorted(unknown)}")
        self._dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[str(dtype)]
        S = weight_standardization
        self.num_classes, self.drop_rate, self.drop_connect_rate, self.seed = num_classes, float(drop_rate), float(drop_connect_rate), int(seed or 0)
        self._seed_given = seed is not None
        self.conv1 = nn.Sequential(_Conv(3, 32, 3, 2, S), _ABN(32, LEAKY_ACT), _Conv(32, 32, 3, 1, S), _ABN(32, LEAKY_ACT), _Conv(32, 64, 3, 1, S))
        self.bn1 = _ABN(64, LEAKY_ACT)
        cin = 64
        for li, (nb, planes) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), 1):
            blocks = nn.ModuleList()
            for i in range(nb):
                blocks.append(_Bottleneck(cin, planes, 2 if (i == 0 and li > 1) else 1, i == 0, S))
                cin = planes * 4
            setattr(self, f"layer{li}", blocks)
        self.fc = nn.Linear(2048, num_classes)
        self._step = 0
        self.masks = None  # test hook: {"dc": [per block [N] fp32 or None], "do": [N, 2048] fp32 or None} overrides the sampler
        self.reset_parameters()

    def reset_parameters(self, seed=None, gamma=1.41421356):
        """train.py:69-71 `pt.utils.misc.initialize(model, cfg.init_gamma)`: kaiming-style conv init with gain `gamma`
        (fan-out), BN weight 1 / bias 0.  The exact law of the un-vendored helper is unknown: initial
        weights are an input of the parity tests, not a parity claim."""
        g = torch.Generator().manual_seed(self.seed if seed is None else int(seed))
        for m in self.modules():
            if isinstance(m, _Conv):
                fan_out = m.weight.shape[0] * m.weight.shape[2] * m.weight.shape[3]
                with torch.no_grad():
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * float(gamma) / fan_out ** 0.5)
            elif isinstance(m, _ABN):
                with torch.no_grad():
                    m.weight.fill_(1.0)
                    m.bias.zero_()
            elif isinstance(m, _ECA):
                with torch.no_grad():
                    m.conv.weight.copy_((torch.rand(m.conv.weight.shape, generator=g) * 2 - 1) * (1.0 / 3.0) ** 0.5)
        with torch.no_grad():
            self.fc.weight.copy_((torch.rand(self.fc.weight.shape, generator=g) * 2 - 1) / 2048 ** 0.5)
            self.fc.bias.zero_()

    def blocks(self):
        return [b for li in range(1, 5) for b in getattr(self, f"layer{li}")]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("bresnet50: the MI355X hot path has no CPU fallback — move the model and the batch to CUDA")
        if not (x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            raise ValueError("bresnet50 expects a CUDA float32 NCHW batch [N,3,H,W] (dali_dataloader.py:113-122 contract)")
        N = x.shape[0]
        h = torch.zeros((N, x.shape[2], x.shape[3], 64), dtype=self._dtype, device=x.device)
        h[..., :3] = x.permute(0, 2, 3, 1)  # the loader's NCHW batch -> zero-padded NHWC
        st = self.conv1
        h = _conv_bn(st[0], st[1], h)
        h = _conv_bn(st[2], st[3], h)
        h = _conv_bn(st[4], self.bn1, h)
        h = _BlurPoolFn.apply(_MaxPool3s1Fn.apply(h))
        blocks = self.blocks()
        train = self.training and torch.is_grad_enabled()
        for i, b in enumerate(blocks):
            keep = None
            if self.masks is not None:
                keep = self.masks["dc"][i]
            elif train and self.drop_connect_rate > 0 and i > 0:
                keep = ops.keep_scale(N, self.drop_connect_rate * i / len(blocks), self.seed, self._step * 64 + i, x.device)
            h = b(h, keep)
        p = _GapFn.apply(h)
        if self.masks is not None:
            if self.masks.get("do") is not None:
                p = p * self.masks["do"]
        elif train and self.drop_rate > 0:
            p = p * ops.keep_scale(p.numel(), self.drop_rate, self.seed, self._step * 64 + 63, x.device).view_as(p)
        if train:
            self._step += 1
        return torch.nn.functional.linear(p, self.fc.weight, self.fc.bias)


_DT = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "torch.bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
       "torch.float32": torch.float32, "None": torch.float32}  # (keys are str(dtype): None = fp32, as models.resnet50)


def _layout(dtype_code, num_classes, wstd):
    """tensor table of the native BResNet-50 executor: [(name, kind, offset, shape)] + flat sizes (layout-only ctx: no GPU needed)"""
    from . import native

    L = native.lib()
    ctx = ctypes.c_void_p()
    native.check(L.mi355_bresnet50_create(ctypes.byref(ctx), -1, dtype_code, 1, 32, 32, num_classes, int(wstd)))
    try:
        table = []
        for i in range(L.mi355_bresnet50_num_tensors(ctx)):
            name = ctypes.create_string_buffer(128)
            kind, off, nd, sh = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_int(), (ctypes.c_int * 4)()
            native.check(L.mi355_bresnet50_tensor_info(ctx, i, name, 128, ctypes.byref(kind), ctypes.byref(off), ctypes.byref(nd), sh))
            table.append((name.value.decode(), kind.value, off.value, tuple(sh[j] for j in range(nd.value))))
        segs = []
        for i in range(L.mi355_bresnet50_num_segments(ctx)):
            b, e = ctypes.c_size_t(), ctypes.c_size_t()
            native.check(L.mi355_bresnet50_segment_range(ctx, i, ctypes.byref(b), ctypes.byref(e)))
            segs.append((b.value, e.value))
        return table, L.mi355_bresnet50_flat_param_elems(ctx), L.mi355_bresnet50_flat_buffer_elems(ctx), segs
    finally:
        L.mi355_bresnet50_destroy(ctx)


class BResNet50(_DropStream, _FlatModel):
    """BResNet-50 on the static executor (csrc/bresnet_exec.cpp): forward and backward are ONE C-ABI call each; every parameter
    (pytorch_tools names / shapes, see the module docstring) is a view into one flat fp32 array, so the native SGD and the flat
    gradient all-reduce apply as they do to models.ResNet50.  No CPU path: a CPU tensor raises."""

    def __init__(self, num_classes=1000, dtype="bf16", drop_rate=0.0, drop_connect_rate=0.0, weight_standardization=False, seed=None, **kw):
        super().__init__()
        unknown = set(kw) - {"pretrained", "stem_type", "antialias", "attn_type", "norm_layer", "norm_act"}
        if unknown:
            raise TypeError(f"bresnet50: unsupported arguments {sorted(unknown)}")
        from . import native

        if str(dtype) not in _DT:
            raise ValueError(f"bresnet50: dtype {dtype!r} (bf16 | fp32; the fp8 step exists for the torchvision-layout resnet50 only)")
        self.compute_dtype = _DT[str(dtype)]
        self._dt = native.dtype_code(self.compute_dtype)
        self.num_classes, self.drop_rate, self.drop_connect_rate, self.seed = int(num_classes), float(drop_rate), float(drop_connect_rate), int(seed or 0)
        self._seed_given = seed is not None
        self.weight_standardization = bool(weight_standardization)
        self._table, self._nparam, self._nbuf, self._segments = _layout(self._dt, self.num_classes, self.weight_standardization)
        # (backward segments of the ONE native backward call: head, bottlenecks last to first, stem — descending through the flat array)
        self._comm = None  # (mi355_comm*, bucket cap in MiB) once a native communicator is attached: the all-reduces run inside the call
        self._flat_params = torch.zeros(self._nparam, dtype=torch.float32)
        self._flat_grads = torch.zeros(self._nparam, dtype=torch.float32)
        self._flat_buffers = torch.zeros(self._nbuf, dtype=torch.float32)
        self._hook = torch.zeros(1, requires_grad=True)
        self._ctxs = OrderedDict()
        self._grads_dirty = False
        self._grad_sync = None  # parallel.FlatBucketDDP: callable(segment, begin, end), run after the backward call
        self._grad_sync_points = None
        self._sync_grads = True
        self._bn_leaves = []
        self._step = 0
        self.masks = None  # test hook: {"dc": [per block [N] fp32 or None], "do": [N, 2048] fp32 or None} overrides the sampler
        self._build_modules()
        self._rebind_views()
        self.reset_parameters()

    def _canonical_order(self):
        return list(self._table)  # the executor registers in pytorch_tools' module order

    def reset_parameters(self, seed=None, gamma=1.41421356):
        """train.py:69-71 `pt.utils.misc.initialize(model, cfg.init_gamma)`: kaiming-style conv init with gain `gamma` (fan-out),
        BN weight 1 / bias 0; the exact law of the un-vendored helper is unknown — initial weights are an
        input of the parity tests, not a parity claim.  Same draws, in the same order, as BResNet50Graph.reset_parameters."""
        g = torch.Generator().manual_seed(self.seed if seed is None else int(seed))
        with torch.no_grad():
            for leaf, attr, kind, off, shape in self._entries:
                name = attr
                if kind != 0:
                    tgt = leaf._buffers[attr]
                    tgt.fill_(1.0 if name == "running_var" else 0.0)
                    continue
                p = leaf._parameters[attr]
                if len(shape) == 4:
                    p.copy_((torch.randn(shape, generator=g) * float(gamma) / (shape[0] * shape[2] * shape[3]) ** 0.5).to(p.device))
                elif len(shape) == 3:
                    p.copy_(((torch.rand(shape, generator=g) * 2 - 1) * (1.0 / 3.0) ** 0.5).to(p.device))
                elif len(shape) == 2:
                    p.copy_(((torch.rand(shape, generator=g) * 2 - 1) / 2048 ** 0.5).to(p.device))
                else:
                    p.fill_(1.0 if (attr == "weight" and isinstance(leaf, _BNLeaf)) else 0.0)

    # ---- native contexts ---------------------------------------------------------------------------------------------
    def _destroy_ctxs(self):
        if self._ctxs:
            from . import native

            L = native.lib()
            for c in self._ctxs.values():
                L.mi355_bresnet50_destroy(c)
            self._ctxs.clear()

    def _ctx(self, N, H, W):
        from . import native

        key = (N, H, W)
        c = self._ctxs.get(key)
        if c is None:
            if not self._flat_params.is_cuda:
                raise RuntimeError("bresnet50: the MI355X hot path has no CPU fallback — call .cuda() first")
            if len(self._ctxs) >= 2:  # train + val batch shapes (each holds every activation AND every gradient: 50 GB at 256 x 224 px)
                pending = getattr(self, "_last", None)
                for k in list(self._ctxs):  # oldest first; the context a pending backward will use is never the one to go
                    if pending is None or self._ctxs[k] is not pending[0]:
                        native.lib().mi355_bresnet50_destroy(self._ctxs.pop(k))
                        break
            L = native.lib()
            c = ctypes.c_void_p()
            dev = self._flat_params.device.index or 0
            native.check(L.mi355_bresnet50_create(ctypes.byref(c), dev, self._dt, N, H, W, self.num_classes, int(self.weight_standardization)))
            native.check(L.mi355_bresnet50_bind(c, native.ptr(self._flat_params), native.ptr(self._flat_grads), native.ptr(self._flat_buffers)))
            if self._comm is not None:
                native.check(L.mi355_bresnet50_set_comm(c, self._comm[0], float(self._comm[1])))
            if not self._sync_grads:
                native.check(L.mi355_bresnet50_set_grad_sync(c, 0))
            self._ctxs[key] = c
        else:
            self._ctxs.move_to_end(key)
        return c

    def _native_forward(self, x, training):
        from . import native

        if not x.is_cuda:
            raise RuntimeError("bresnet50: the MI355X hot path has no CPU fallback — move the model and the batch to CUDA")
        if not (x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            raise ValueError("bresnet50 expects a CUDA float32 NCHW batch [N,3,H,W] (dali_dataloader.py:113-122 contract)")
        x = x.contiguous()
        N, _, H, W = x.shape
        c = self._ctx(N, H, W)
        L = native.lib()
        # draw drop-connect / dropout on the device: decided by `training` (True only on the autograd training path — grad mode itself
        # reads False inside autograd.Function.forward, so it cannot be asked here)
        sample = bool(training and self.training and self.masks is None)
        native.check(L.mi355_bresnet50_set_drop(c, self.drop_rate, self.drop_connect_rate, self.seed))
        keep_arr, do_ptr, alive = None, None, []
        if not sample:  # given masks (test hook), or none at all (eval / no_grad): the generator stays off
            keep_arr = (ctypes.c_void_p * 16)()
            if self.masks is not None:
                for i, k in enumerate(self.masks["dc"]):
                    if k is not None:
                        k = k.to(device=x.device, dtype=torch.float32).contiguous()
                        alive.append(k)
                        keep_arr[i] = k.data_ptr()
                d = self.masks.get("do")
                if d is not None:
                    d = d.to(device=x.device, dtype=torch.float32).contiguous()
                    alive.append(d)
                    do_ptr = d.data_ptr()
        logits = torch.empty((N, self.num_classes), dtype=torch.float32, device=x.device)
        native.check(L.mi355_bresnet50_forward(c, native.ptr(x), native.ptr(logits), int(bool(self.training)), self.bn_momentum(), self._step, keep_arr,
                                               do_ptr, native.cur_stream()))
        if training:
            self._last = (c, x, alive)  # keep the input (and the mask tensors: copied on the stream) alive until backward
        else:
            self._eval_alive = (x, alive)  # (an eval forward between a training forward and its backward leaves that pair alone)
        if self.training:
            self._nbt += 1
            if training:
                self._step += 1
        return logits

    def _native_backward(self, dlogits):
        from . import native

        if getattr(self, "_last", None) is None:
            raise RuntimeError("bresnet50: backward without a pending training forward")
        c = self._last[0]
        self._attach_grads()
        native.check(native.lib().mi355_bresnet50_backward(c, native.ptr(dlogits.contiguous()), int(self._grads_dirty), native.cur_stream()))
        self._last = None  # the context may be evicted again
        if self._grad_sync is not None and self._sync_grads and self._comm is None:  # (torch.distributed stand-in: after the one call)
            for k, (b, e) in enumerate(self._segments):
                if self._grad_sync_points is None or k in self._grad_sync_points:
                    self._grad_sync(k, b, e)
        self._grads_dirty = True

    def grad_hooks(self, record=None, replay=None):
        """test hook (mi355_bresnet50_grad_hooks) of the next backward of the pending training forward: record / replay are lists of 16 tensors (or None
        entries) shaped like the blocks' outputs in the compute dtype — the gradient each block's backward starts from is copied out / replaced"""
        from . import native

        if getattr(self, "_last", None) is None:
            raise RuntimeError("bresnet50: grad_hooks without a pending training forward")
        arrs = []
        for lst in (record, replay):
            a = (ctypes.c_void_p * 16)()
            for i, t in enumerate(lst or []):
                if t is not None:
                    a[i] = t.data_ptr()
            arrs.append(a)
        self._hook_alive = (record, replay)
        native.check(native.lib().mi355_bresnet50_grad_hooks(self._last[0], arrs[0], arrs[1], 16))

    def debug_tensor(self, shape, name):
        """copy of an internal tensor of the last forward at batch shape (N,H,W) — test hook (mi355_bresnet50_debug_tensor)."""
        from . import native

        L = native.lib()
        p, dt, nd, sh = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), (ctypes.c_int * 4)()
        native.check(L.mi355_bresnet50_debug_tensor(self._ctx(*shape), name.encode(), ctypes.byref(p), ctypes.byref(dt), ctypes.byref(nd), sh))
        dims = [sh[i] for i in range(nd.value)]
        out = torch.empty(dims, dtype={native.F32: torch.float32, native.BF16: torch.bfloat16}[dt.value], device=self._flat_params.device)
        torch.cuda.synchronize()
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        rc = hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), p, out.numel() * out.element_size(), 3)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed ({rc})")
        return out

    def set_comm(self, comm, bucket_cap_mb=32.0):
        """attaches a native RCCL communicator (parallel.FlatBucketDDP owns it): the backward call then reduces the flat gradient array
        bucket by bucket behind the segments that complete it (mi355_bresnet50_set_comm)"""
        from . import native

        self._comm = None if comm is None else (comm, float(bucket_cap_mb))
        for c in self._ctxs.values():
            native.check(native.lib().mi355_bresnet50_set_comm(c, comm, float(bucket_cap_mb)))

    def set_grad_sync(self, on):
        """DDP.no_sync(): off -> backward leaves the gradients rank-local (mi355_bresnet50_set_grad_sync)"""
        from . import native

        self._sync_grads = bool(on)
        for c in self._ctxs.values():
            native.check(native.lib().mi355_bresnet50_set_grad_sync(c, int(self._sync_grads)))

    def bucket_plan(self, bucket_cap_mb):
        """[(begin, end, last_segment)] the native executor would reduce at this cap (layout-only: works on the CPU)"""
        from . import native

        L = native.lib()
        ctx = ctypes.c_void_p()
        native.check(L.mi355_bresnet50_create(ctypes.byref(ctx), -1, self._dt, 1, 32, 32, self.num_classes, int(self.weight_standardization)))
        try:
            n = ctypes.c_int()
            B, E, S = (ctypes.c_size_t * 32)(), (ctypes.c_size_t * 32)(), (ctypes.c_int * 32)()
            native.check(L.mi355_bresnet50_bucket_plan(ctx, float(bucket_cap_mb), 32, ctypes.byref(n), B, E, S))
            return [(B[i], E[i], S[i]) for i in range(n.value)]
        finally:
            L.mi355_bresnet50_destroy(ctx)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return _ResNetFn.apply(x, self._hook, self)
        return self._native_forward(x, training=False)  # eval, or train mode under no_grad: batch statistics as the mode says, no drops

    def flops(self, N, H, W):
        """(forward, training) algorithmic FLOPs of one step at this shape (2 FLOP/MAC, conv + FC)"""
        from . import native

        L = native.lib()
        ctx = ctypes.c_void_p()
        native.check(L.mi355_bresnet50_create(ctypes.byref(ctx), -1, self._dt, N, H, W, self.num_classes, int(self.weight_standardization)))
        f, t = ctypes.c_double(), ctypes.c_double()
        native.check(L.mi355_bresnet50_flops(ctx, ctypes.byref(f), ctypes.byref(t)))
        L.mi355_bresnet50_destroy(ctx)
        return f.value, t.value


def bresnet50(**kwargs):
    """`_target_` plugin for the BResNet-50 recipes (BASELINE configs[3])."""
    return BResNet50(**kwargs)
