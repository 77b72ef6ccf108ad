"""`resnet50` plugin: a torch.nn.Module facade over the native MI355X executor (csrc/resnet_exec.cpp).

Drop-in for the reference's model plugin call `hydra.utils.call(cfg.model)` with
`_target_: pytorch_tools.models.resnet50` (train.py:64; configs/hydra_exp/1.r50_baseline.yaml:22-23):
callable `(data NCHW fp32) -> logits`, `.cuda()`, `.parameters()`, `.state_dict()` /
`.load_state_dict(strict=False)` with torchvision names and shapes (train.py:73,86,101,184), `.modules()`
walk with BatchNorm-like `.momentum` (train.py:76 -> patch_bn_mom), `.train()/.eval()`.

All 161 parameters are views into ONE flat fp32 buffer (gradients likewise), laid out by the native library
in reverse execution order so DDP buckets complete front to back; conv weights are torch `[Cout,Cin,KH,KW]`
tensors in channels_last memory (= the KRSC layout the kernels read).  Nothing here computes: forward and
backward are single C-ABI calls; without the HIP library / a GPU they raise.
"""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from . import native
from .native import check, ptr

_DTYPES = {
    None: torch.float32, "fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
    "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
    torch.float32: torch.float32, torch.bfloat16: torch.bfloat16,
    # BASELINE configs[4] "fp8 MFMA convs": bf16 tensors, e4m3 operand twins for the forward / dgrad convs of layers 2-4
    "fp8": torch.bfloat16, "float8": torch.bfloat16, "e4m3": torch.bfloat16,
}
_FP8_NAMES = ("fp8", "float8", "e4m3")


class _Leaf(nn.Module):
    """container node giving parameters their torchvision names (conv1, bn1, layer1.0.conv1, ...)"""


class _BNLeaf(_Leaf):
    """BatchNorm-like leaf: carries `.momentum` / `.eps` so pt.utils.misc.patch_bn_mom (train.py:76) has something to patch."""

    def __init__(self):
        super().__init__()
        self.momentum = 0.1
        self.eps = 1e-5


def _layout(dtype_code, N, H, W, num_classes):
    """tensor table of the native executor: [(name, kind, offset, shape)], sizes, segment ranges (no GPU needed)."""
    L = native.lib()
    ctx = ctypes.c_void_p()
    check(L.mi355_resnet50_create(ctypes.byref(ctx), -1, dtype_code, N, H, W, num_classes))
    try:
        table = []
        for i in range(L.mi355_resnet50_num_tensors(ctx)):
            name = ctypes.create_string_buffer(128)
            kind, off, nd, sh = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_int(), (ctypes.c_int * 4)()
            check(L.mi355_resnet50_tensor_info(ctx, i, name, 128, ctypes.byref(kind), ctypes.byref(off), ctypes.byref(nd), sh))
            table.append((name.value.decode(), kind.value, off.value, tuple(sh[j] for j in range(nd.value))))
        segs = []
        for s in range(L.mi355_resnet50_num_segments(ctx)):
            b, e = ctypes.c_size_t(), ctypes.c_size_t()
            check(L.mi355_resnet50_segment_range(ctx, s, ctypes.byref(b), ctypes.byref(e)))
            segs.append((b.value, e.value))
        return table, L.mi355_resnet50_flat_param_elems(ctx), L.mi355_resnet50_flat_buffer_elems(ctx), segs
    finally:
        L.mi355_resnet50_destroy(ctx)


class _ResNetFn(torch.autograd.Function):
    """autograd bridge so reference-style `loss.backward()` (callbacks.py:317) drives the native backward."""

    @staticmethod
    def forward(ctx, x, hook, model):
        ctx.model = model
        return model._native_forward(x, training=True)

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._native_backward(dlogits)
        return None, None, None


class _FlatModel(nn.Module):
    """What the facades over the two static executors (ResNet-50 here, BResNet-50 in bresnet.py) share: every parameter / buffer is
    a view into ONE flat fp32 array (gradients likewise) laid out by the native library; subclasses provide `_table`
    [(name, kind, offset, shape)], `_canonical_order()`, the flat arrays, `_segments`, and `_destroy_ctxs()`."""

    def _leaf(self, dotted, bn=False):
        node = self
        parts = dotted.split(".")
        for i, p in enumerate(parts):
            if p not in node._modules:
                last = i == len(parts) - 1
                node.add_module(p, _BNLeaf() if (bn and last) else _Leaf())
            node = node._modules[p]
        return node

    def _build_modules(self):
        self._entries = []  # (leaf, attr, kind, offset, shape)
        for name, kind, off, shape in self._canonical_order():
            mod, attr = name.rsplit(".", 1)
            is_bn = kind == 1 or (len(shape) == 1 and not mod.startswith("fc"))
            leaf = self._leaf(mod, bn=is_bn)
            if is_bn and leaf not in self._bn_leaves:
                self._bn_leaves.append(leaf)
            self._entries.append((leaf, attr, kind, off, shape))
            (leaf._parameters if kind == 0 else leaf._buffers)[attr] = None  # fixes the registration order
        # one int64 array for all 53 counters (a single `+= 1` per step); each leaf's buffer is a 0-dim view of it
        self._nbt = torch.zeros(len(self._bn_leaves), dtype=torch.int64)
        for i, leaf in enumerate(self._bn_leaves):
            leaf.register_buffer("num_batches_tracked", self._nbt[i])

    @staticmethod
    def _view(flat, off, shape):
        n = 1
        for s in shape:
            n *= s
        v = flat[off: off + n]
        if len(shape) == 4:  # torch OIHW logical shape over KRSC memory (channels_last)
            co, ci, kh, kw = shape
            return v.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return v.view(*shape)

    def _rebind_views(self):
        """(re)creates every Parameter / buffer as a view of the flat arrays (after construction or a device move)."""
        for leaf, attr, kind, off, shape in self._entries:
            if kind == 0:
                p = nn.Parameter(self._view(self._flat_params, off, shape), requires_grad=True)
                p.grad = self._view(self._flat_grads, off, shape)
                leaf._parameters[attr] = p
            else:
                leaf._buffers[attr] = self._view(self._flat_buffers, off, shape)

    def _attach_grads(self):
        """(re)binds every parameter's .grad to its slice of the flat gradient array.  A gradient that was set to None
        since the last backward — model.zero_grad(), any torch optimizer's zero_grad(set_to_none=True) — counts as
        zeroed: if all are None the next backward overwrites, otherwise the stale slices of the None ones are cleared."""
        none, total = [], 0
        for leaf, attr, kind, off, shape in self._entries:
            if kind == 0:
                p = leaf._parameters[attr]
                total += 1
                if p.grad is None:
                    none.append((off, shape))
                if p.grad is None or p.grad.data_ptr() != self._flat_grads.data_ptr() + off * 4:
                    p.grad = self._view(self._flat_grads, off, shape)
        if self._grads_dirty and none:
            if len(none) == total:
                self._grads_dirty = False
            else:
                for off, shape in none:
                    self._view(self._flat_grads, off, shape).zero_()

    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad: the flat gradient array is simply overwritten by the next backward."""
        super().zero_grad(set_to_none=set_to_none)
        self.mark_grads_clean()

    def _apply(self, fn, recurse=True):
        # move the flat arrays, then re-create the views; native contexts belong to the old device
        self._destroy_ctxs()
        self._flat_params = fn(self._flat_params)
        self._flat_grads = fn(self._flat_grads)
        self._flat_buffers = fn(self._flat_buffers)
        self._hook = fn(self._hook.detach()).requires_grad_(True)
        self._nbt = fn(self._nbt)
        for i, leaf in enumerate(self._bn_leaves):
            leaf._buffers["num_batches_tracked"] = self._nbt[i]
        if self._flat_params.dtype != torch.float32:
            raise TypeError("master parameters stay fp32; choose the compute dtype with resnet50(dtype='bf16')")
        self._rebind_views()
        return self

    def _name_of(self, leaf):
        for n, m in self.named_modules():
            if m is leaf:
                return n
        raise KeyError

    # ---- flat access (optimizer / DDP) ---------------------------------------------------------------------
    @property
    def flat_params(self):
        return self._flat_params

    @property
    def flat_grads(self):
        return self._flat_grads

    @property
    def grad_segments(self):
        """[(begin, end)] element ranges of the flat gradient array, in backward completion order."""
        return list(self._segments)

    def mark_grads_clean(self):
        """the next backward overwrites the flat gradients instead of accumulating (optimizer.zero_grad())."""
        self._grads_dirty = False

    def __del__(self):
        try:
            self._destroy_ctxs()
        except Exception:
            pass

    def bn_momentum(self):
        return float(self._bn_leaves[0].momentum) if self._bn_leaves else 0.1


class ResNet50(_FlatModel):
    def __init__(self, num_classes=1000, dtype=None, pretrained=None, **unsupported):
        super().__init__()
        if pretrained:
            raise ValueError("pretrained weights are not available offline")
        # architecture kwargs of pytorch_tools.models.resnet50 that would change the graph are rejected loudly
        for k, v in unsupported.items():
            if v not in (None, False, "", 0, 0.0, "relu", "abn"):
                raise NotImplementedError(f"resnet50({k}={v!r}) is outside the MI355X hot path (SURVEY.md §8f)")
        self.num_classes = int(num_classes)
        self.compute_dtype = _DTYPES[dtype]
        self.fp8 = isinstance(dtype, str) and dtype in _FP8_NAMES
        self._dt = native.FP8 if self.fp8 else native.dtype_code(self.compute_dtype)
        table, self._nparam, self._nbuf, self._segments = _layout(self._dt, 1, 32, 32, self.num_classes)
        self._table = table
        self._flat_params = torch.zeros(self._nparam, dtype=torch.float32)
        self._flat_grads = torch.zeros(self._nparam, dtype=torch.float32)
        self._flat_buffers = torch.zeros(self._nbuf, dtype=torch.float32)
        self._hook = torch.zeros(1, requires_grad=True)  # gives autograd an edge into _ResNetFn
        self._ctxs = OrderedDict()  # (N,H,W) -> native ctx
        self._grads_dirty = False
        self._grad_sync = None  # set by parallel.FlatBucketDDP: callable(segment, begin, end)
        self._grad_sync_points = None  # optional set of segments the hook acts on (None: after every segment)
        self._comm = None  # (mi355_comm*, bucket cap in MiB) once a native communicator is attached
        self._sync_grads = True  # False inside FlatBucketDDP.no_sync(): backward keeps the gradients rank-local
        self._bn_leaves = []
        self._build_modules()
        self._rebind_views()
        self.reset_parameters()

    # ---- module tree with torchvision names --------------------------------------------------------------
    def _canonical_order(self):
        """torchvision registration order: conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var."""
        by_name = {t[0]: t for t in self._table}
        convs = [n[: -len(".weight")] for n, k, _, sh in self._table if k == 0 and len(sh) == 4]
        order = []

        def convbn(conv, bn):
            order.append(by_name[conv + ".weight"])
            for suffix in ("weight", "bias", "running_mean", "running_var"):
                order.append(by_name[f"{bn}.{suffix}"])

        convbn("conv1", "bn1")
        blocks = sorted({c.rsplit(".", 1)[0] for c in convs if c.startswith("layer") and "downsample" not in c},
                        key=lambda s: (int(s[5]), int(s.split(".")[1])))
        for b in blocks:
            for i in (1, 2, 3):
                convbn(f"{b}.conv{i}", f"{b}.bn{i}")
            if f"{b}.downsample.0.weight" in by_name:
                convbn(f"{b}.downsample.0", f"{b}.downsample.1")
        order.append(by_name["fc.weight"])
        order.append(by_name["fc.bias"])
        assert len(order) == len(self._table)
        return order

    def reset_parameters(self, seed=0, gamma=1.72):
        from .synth import init_state_dict

        shapes = [(f"{self._name_of(leaf)}.{attr}", shape) for leaf, attr, kind, off, shape in self._entries]
        sd = init_state_dict(shapes, seed=seed, gamma=gamma)
        with torch.no_grad():
            for (leaf, attr, kind, off, shape), (name, _) in zip(self._entries, shapes):
                tgt = leaf._parameters[attr] if kind == 0 else leaf._buffers[attr]
                tgt.copy_(sd[name].to(tgt.device))

    def set_comm(self, comm, bucket_cap_mb=32.0):
        """attaches a native RCCL communicator (mi355_comm*, parallel.FlatBucketDDP owns it): every backward then reduces
        the flat gradient array bucket by bucket inside the ONE native backward call (mi355_resnet50_set_comm)."""
        self._comm = None if comm is None else (comm, float(bucket_cap_mb))
        L = native.lib()
        for c in self._ctxs.values():
            check(L.mi355_resnet50_set_comm(c, comm, float(bucket_cap_mb)))

    def set_grad_sync(self, on):
        """DDP.no_sync() for the native collective: off -> the following backwards skip the bucket all-reduces
        (mi355_resnet50_set_grad_sync); the torch.distributed stand-in path honours the same flag."""
        self._sync_grads = bool(on)
        L = native.lib()
        for c in self._ctxs.values():
            check(L.mi355_resnet50_set_grad_sync(c, int(self._sync_grads)))

    def bucket_plan(self, bucket_cap_mb):
        """[(begin, end, last_segment)] the native executor would reduce at this cap (layout-only: works on the CPU)."""
        L = native.lib()
        ctx = ctypes.c_void_p()
        check(L.mi355_resnet50_create(ctypes.byref(ctx), -1, self._dt, 1, 32, 32, self.num_classes))
        try:
            n = ctypes.c_int()
            B, E, S = (ctypes.c_size_t * 32)(), (ctypes.c_size_t * 32)(), (ctypes.c_int * 32)()
            check(L.mi355_resnet50_bucket_plan(ctx, float(bucket_cap_mb), 32, ctypes.byref(n), B, E, S))
            return [(B[i], E[i], S[i]) for i in range(n.value)]
        finally:
            L.mi355_resnet50_destroy(ctx)

    # ---- native contexts -------------------------------------------------------------------------------------
    def _destroy_ctxs(self):
        if self._ctxs:
            L = native.lib()
            for c in self._ctxs.values():
                L.mi355_resnet50_destroy(c)
            self._ctxs.clear()

    def _ctx(self, N, H, W):
        key = (N, H, W)
        c = self._ctxs.get(key)
        if c is None:
            if not self._flat_params.is_cuda:
                raise RuntimeError("resnet50: the MI355X hot path has no CPU fallback — call .cuda() first")
            if len(self._ctxs) >= 3:  # progressive resize / val batch: keep the 3 most recent shapes
                _, old = self._ctxs.popitem(last=False)
                native.lib().mi355_resnet50_destroy(old)
            L = native.lib()
            c = ctypes.c_void_p()
            dev = self._flat_params.device.index or 0
            check(L.mi355_resnet50_create(ctypes.byref(c), dev, self._dt, N, H, W, self.num_classes))
            check(L.mi355_resnet50_bind(c, ptr(self._flat_params), ptr(self._flat_grads), ptr(self._flat_buffers)))
            if self._comm is not None:
                check(L.mi355_resnet50_set_comm(c, self._comm[0], float(self._comm[1])))
            if not self._sync_grads:
                check(L.mi355_resnet50_set_grad_sync(c, 0))
            self._ctxs[key] = c
        else:
            self._ctxs.move_to_end(key)
        return c

    def _native_forward(self, x, training):
        if not x.is_cuda:
            raise RuntimeError("resnet50: the MI355X hot path has no CPU fallback — move the model and the batch to CUDA")
        if not (x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            raise ValueError("resnet50 expects a CUDA float32 NCHW batch [N,3,H,W] (dali_dataloader.py:113-122 contract)")
        x = x.contiguous()
        N, _, H, W = x.shape
        c = self._ctx(N, H, W)
        logits = torch.empty((N, self.num_classes), dtype=torch.float32, device=x.device)
        check(native.lib().mi355_resnet50_forward(c, ptr(x), ptr(logits), int(training), self.bn_momentum(), native.cur_stream()))
        self._last = (c, x)  # keep the input alive until backward
        if training:
            self._nbt += 1
        return logits

    def _native_backward(self, dlogits):
        c, _ = self._last
        L = native.lib()
        dlogits = dlogits.contiguous()
        self._attach_grads()
        acc = int(self._grads_dirty)
        nseg = len(self._segments)
        if self._grad_sync is None or not self._sync_grads:
            check(L.mi355_resnet50_backward(c, ptr(dlogits), 0, nseg, acc, native.cur_stream()))
        else:
            # one native call per run of segments up to the next segment the hook acts on (a bucket boundary): every
            # call ends by joining the weight-gradient side stream, so fewer calls keep more of that overlap
            points = self._grad_sync_points
            s0 = 0
            for s in range(nseg):
                if points is None or s in points or s == nseg - 1:
                    check(L.mi355_resnet50_backward(c, ptr(dlogits), s0, s + 1, acc, native.cur_stream()))
                    for k in range(s0, s + 1):
                        self._grad_sync(k, *self._segments[k])
                    s0 = s + 1
        self._grads_dirty = True

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return _ResNetFn.apply(x, self._hook, self)
        return self._native_forward(x, training=self.training)

    def flops(self, N, H, W):
        """(forward, training) algorithmic FLOPs of one step at this shape (2 FLOP/MAC, conv + FC)."""
        L = native.lib()
        ctx = ctypes.c_void_p()
        check(L.mi355_resnet50_create(ctypes.byref(ctx), -1, self._dt, N, H, W, self.num_classes))
        f, t = ctypes.c_double(), ctypes.c_double()
        check(L.mi355_resnet50_flops(ctx, ctypes.byref(f), ctypes.byref(t)))
        L.mi355_resnet50_destroy(ctx)
        return f.value, t.value

    def kernel_table(self, shape):
        """{conv name: {"fwd": kernel, "dgrad": kernel, "wgrad": kernel}} of the last step at batch shape (N,H,W) — which kernel each
        convolution's launches went to (mi355_resnet50_kernel_table; '-' = not launched)."""
        L = native.lib()
        need = ctypes.c_size_t(0)
        check(L.mi355_resnet50_kernel_table(self._ctx(*shape), None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        check(L.mi355_resnet50_kernel_table(self._ctx(*shape), buf, need.value, None))
        out = {}
        for ln in buf.value.decode().splitlines():
            name, *kv = ln.split()
            out[name] = dict(x.split("=", 1) for x in kv)
        return out

    def debug_tensor(self, shape, name):
        """copy of an internal tensor of the last step at batch shape (N,H,W) — test hook (mi355_resnet50_debug_tensor)."""
        L = native.lib()
        p, dt, nd, sh = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), (ctypes.c_int * 4)()
        check(L.mi355_resnet50_debug_tensor(self._ctx(*shape), name.encode(), ctypes.byref(p), ctypes.byref(dt), ctypes.byref(nd), sh))
        dims = [sh[i] for i in range(nd.value)]
        tdt = {native.F32: torch.float32, native.BF16: torch.bfloat16, native.FP8: torch.uint8}[dt.value]  # FP8: raw e4m3 bytes
        out = torch.empty(dims, dtype=tdt, device=self._flat_params.device)
        torch.cuda.synchronize()
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        rc = hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), p, out.numel() * out.element_size(), 3)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed ({rc})")
        return out

    def force_grad(self, shape, g):
        """teacher forcing between two backward segments: the next segment starts from `g` instead of the gradient the previous one left
        (test hook, mi355_resnet50_force_grad)"""
        g = g.contiguous()
        check(native.lib().mi355_resnet50_force_grad(self._ctx(*shape), ptr(g), g.numel() * g.element_size(), native.cur_stream()))

    def fp8_state(self, shape):
        """(forward twins in use, gradient twins in use, fp8 forward layers, fp8 dgrad layers) of the last training step at
        batch shape (N,H,W) — mi355_resnet50_fp8_state"""
        v = [ctypes.c_int() for _ in range(4)]
        check(native.lib().mi355_resnet50_fp8_state(self._ctx(*shape), *[ctypes.byref(x) for x in v]))
        return bool(v[0].value), bool(v[1].value), v[2].value, v[3].value

    # profiling passthrough (bench.py)
    def profile(self, shape, class_mask):
        check(native.lib().mi355_resnet50_profile(self._ctx(*shape), int(class_mask)))

    def profile_read(self, shape, kind):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        check(native.lib().mi355_resnet50_profile_read(self._ctx(*shape), kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
        return ms.value, n.value, fl.value, by.value


_VARIANT_KEYS = ("stem_type", "antialias", "attn_type", "norm_layer", "norm_act", "drop_rate", "drop_connect_rate", "weight_standardization")


def _is_variant(kw):
    return any(kw.get(k) not in (None, False, "", 0, 0.0, "relu", "abn") for k in _VARIANT_KEYS)


def _bresnet50(**kw):
    """pytorch_tools.models.resnet50 with the BResNet-50 model_params (BResNet50_encoder.yaml:41-51): the one variant graph
    that is built (deep stem, anti-aliasing, ECA, leaky-ReLU ABN, drop-connect / dropout, optional weight standardisation)."""
    from .bresnet import BResNet50

    want = dict(stem_type="deep", antialias=True, attn_type="eca", norm_act="leaky_relu")
    for k, v in want.items():
        if kw.get(k) != v:
            raise NotImplementedError(f"resnet50({k}={kw.get(k)!r}): the variant graph on the MI355X path is the BResNet-50 of BASELINE configs[3] "
                                      f"({', '.join(f'{a}={b!r}' for a, b in want.items())}, norm_layer inplaceabn|abn)")
    if kw.get("norm_layer") not in ("inplaceabn", "abn", None):
        raise NotImplementedError(f"resnet50(norm_layer={kw.get('norm_layer')!r}) is outside the MI355X hot path")
    return BResNet50(**kw)


def resnet50(**kwargs):
    """plugin entry point — same name as the reference's `_target_: pytorch_tools.models.resnet50`.  Plain kwargs give the
    torchvision-layout ResNet-50 on the static executor; the BResNet-50 model_params give the variant graph (bresnet.py)."""
    if _is_variant(kwargs):
        return _bresnet50(**kwargs)
    return ResNet50(**kwargs)
