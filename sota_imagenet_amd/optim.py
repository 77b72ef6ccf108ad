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
        self._ema = None       # (flat parameter array, its moving average, decay): attach_ema()

    def attach_model(self, model):
        """lets zero_grad() tell the model that the next backward may overwrite its flat gradients, and the range
        planner see every tensor of the model (which gaps between updated parameters are padding)."""
        self._models.append(model)
        self._plans = None

    def attach_ema(self, flat_params, flat_ema, decay):
        """the moving average of a model's flat parameter array (fit_wrapper.ModelEma, train.py:111-112) is advanced by the step kernel
        itself: ema += (1 - decay) * (p_new - ema) over every range this optimizer updates (ranges it does not update never change, so
        their average stays what it was cloned from).  detach_ema() hands the job back to the callback."""
        if not (flat_ema.is_cuda and flat_ema.dtype == torch.float32 and flat_ema.is_contiguous() and flat_ema.numel() == flat_params.numel()):
            raise ValueError("attach_ema: the average must be a contiguous CUDA fp32 tensor of the flat array's size")
        self._ema = (flat_params, flat_ema, float(decay))
        self._plans = None

    def detach_ema(self):
        self._ema = None
        self._plans = None

    # one plan per param group: list of (p_flat_slice, g_flat_slice, m_flat_slice, ema_flat_slice or None)
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
        # Neighbours of one group merge into one launch only when the gap between them is PROVABLY padding (zeros stay
        # zeros under the update): every tensor of the attached models that this optimizer does not update in the same
        # group — frozen parameters, parameters of another group, parameters without a gradient — acts as a barrier.
        # Without an attached model nothing is known about a gap, so only the 64-element alignment padding is bridged.
        entries.sort(key=lambda r: (r[0], r[2]))
        mine = {id(e[5]) for e in entries}
        barriers = {}  # storage base -> sorted [(first elem, end elem)] of tensors that must not be swept up
        for mdl in self._models:
            for q in mdl.parameters():
                if id(q) in mine:
                    continue
                r = _dense_range(q.data)
                if r is not None:
                    barriers.setdefault(r[0], []).append((r[1], r[1] + r[2]))
        max_gap = 64 * 2048 if self._models else 64

        def gap_is_padding(pb, lo, hi):
            return 0 <= hi - lo < max_gap and not any(b < hi and e > lo for b, e in barriers.get(pb, ()))

        merged = []
        for pb, gb, off, n, gi, p in entries:
            m = merged[-1] if merged else None
            if m and m[0] == pb and m[1] == gb and m[5] == gi and gap_is_padding(pb, m[3], off):
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
                view = torch.as_strided(fm, p.shape, p.stride(), r[1] - b)
                old = self.state[p].get("momentum_buffer")
                if old is not None:  # loaded from a checkpoint (train.py:144) or kept across a re-plan: carry it over
                    view.copy_(old.to(device=dev, dtype=torch.float32))
                self.state[p]["momentum_buffer"] = view
            fe = None
            if self._ema is not None:
                r = _dense_range(self._ema[0])
                if r is not None and r[0] == pb and r[1] <= b and e <= r[1] + r[2]:
                    fe = self._ema[1][b - r[1]: e - r[1]]
            plans[gi].append((fp, fg, fm, fe))
        if self._ema is not None and not any(fe is not None for segs in plans for *_, fe in segs):
            raise RuntimeError("SGD.attach_ema: none of the updated ranges lies in the attached flat array")
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
            for fp, fg, fm, fe in segs:
                ops.sgd_step(fp, fg, fm, float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]), float(self.grad_scale),
                             ema=fe, ema_decay=self._ema[2] if fe is not None else 0.0)
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

    def load_state_dict(self, state_dict):
        """torch's loader replaces self.state[p]['momentum_buffer'] by fresh tensors: re-plan at the next step, which
        copies them into the flat momentum array the kernel reads (resume path, train.py:140-146)."""
        super().load_state_dict(state_dict)
        self._plans = None
