"""`SGD` plugin: torch.optim-compatible SGD with momentum whose step is ONE fused HIP kernel over the flat buffer.

Drop-in for `_target_: torch.optim._multi_tensor.SGD` (sota_imagenet/arg_parser.py:136-138; r50 recipe adds
momentum 0.9 / weight_decay 3e-5, configs/hydra_exp/1.r50_baseline.yaml:29-31; built at train.py:92 from
`opt_params = [{"params": [...]}(, {"params": [...], "weight_decay": 0})]`, train.py:83-89; the scheduler writes
`param_group["lr"]` every batch).  Semantics = torch.optim.SGD with dampening 0, nesterov off:
    g += wd * p ;  m = mu * m + g  (first step m = g) ;  p -= lr * m
Parameters that are views of a model's flat fp32 array (models.ResNet50) are updated range-wise in place;
adjacent ranges of one param group collapse into a single launch (the default recipe = 1 launch / step).
"""
import torch
from torch.optim import Optimizer

from . import ops


def _dense_range(t):
    """(storage base ptr, first elem, numel) if `t` covers a dense memory range (any permutation of strides)."""
    n = t.numel()
    sizes_strides = sorted(zip(t.stride(), t.size()))
    expect = 1
    for st, sz in sizes_strides:
        if sz == 1:
            continue
        if st != expect:
            return None
        expect *= sz
    base = t.untyped_storage().data_ptr()
    return base, (t.data_ptr() - base) // t.element_size(), n


class SGD(Optimizer):
    def __init__(self, params, lr=0.0, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, **ignored):
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("dampening / nesterov are not on the hot path")
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._plans = None
        self._models = []
        self.grad_scale = 1.0  # e.g. 1/world_size when gradients were summed, not averaged

    def attach_model(self, model):
        """lets zero_grad() tell the model that the next backward may overwrite its flat gradients."""
        self._models.append(model)

    # one plan per param group: list of (p_flat_slice, g_flat_slice, m_flat_slice)
    def _build_plans(self):
        entries = []  # (param base, grad base, first elem, numel, group index, param)
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32):
                    raise RuntimeError("SGD: parameters must be CUDA fp32 tensors (no CPU fallback on the hot path)")
                rp, rg = _dense_range(p.data), _dense_range(p.grad)
                if rp is None or rg is None or rp[1:] != rg[1:]:
                    raise RuntimeError("SGD: parameter and gradient must be dense and share their flat offset")
                entries.append((rp[0], rg[0], rp[1], rp[2], gi, p))
        # walk ALL parameters in memory order; neighbours merge only when they belong to the same group, so a gap
        # that is merged over can hold nothing but alignment / FC row padding (zeros stay zeros under the update)
        entries.sort(key=lambda r: (r[0], r[2]))
        merged = []
        for pb, gb, off, n, gi, p in entries:
            m = merged[-1] if merged else None
            if m and m[0] == pb and m[1] == gb and m[5] == gi and 0 <= off - m[3] < 64 * 2048:
                m[3] = off + n
                m[4].append(p)
            else:
                merged.append([pb, gb, off, off + n, [p], gi])
        plans = [[] for _ in self.param_groups]
        for pb, gb, b, e, ps, gi in merged:
            if b * 4 % 16:
                raise RuntimeError("SGD: flat range not 16-byte aligned")
            dev = ps[0].device
            fp = torch.empty(0, dtype=torch.float32, device=dev).set_(ps[0].data.untyped_storage(), b, (e - b,))
            fg = torch.empty(0, dtype=torch.float32, device=dev).set_(ps[0].grad.untyped_storage(), b, (e - b,))
            fm = torch.zeros(e - b, dtype=torch.float32, device=dev)
            for p in ps:  # expose momentum buffers per parameter (state_dict compatibility)
                r = _dense_range(p.data)
                self.state[p]["momentum_buffer"] = torch.as_strided(fm, p.shape, p.stride(), r[1] - b)
            plans[gi].append((fp, fg, fm))
        self._plans = plans

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._plans is None:
            self._build_plans()
        for group, segs in zip(self.param_groups, self._plans):
            for fp, fg, fm in segs:
                ops.sgd_step(fp, fg, fm, float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]), float(self.grad_scale))
        return loss

    def zero_grad(self, set_to_none=False):
        # gradients live in the model's flat array and are overwritten by the next backward: no memset needed
        if self._models:
            for m in self._models:
                m.mark_grads_clean()
        else:
            super().zero_grad(set_to_none=False)

    def add_param_group(self, group):
        super().add_param_group(group)
        self._plans = None
