"""Input-side callbacks next to the hot path: Mixup / CutMix / CutmixMixup on device, producing the soft targets the
native cross entropy consumes.

Re-states sota_imagenet/callbacks.py:232-247 (`CutmixMixup`: coin flip between `self.cutmix(*input)` and
`self.mixup(*input)` with Beta(alpha, alpha) samplers) and the un-vendored pt_clb.Cutmix / pt_clb.Mixup bases as
SURVEY.md Appendix C records them (mix with the PREVIOUS batch, permuted; CutMix target weight = real box area).
"""

import numpy as np
import torch

from .fit_wrapper import Callback


class _DeviceMixer:
    """the native path (csrc/mix.hip through the C-ABI): decisions sampled ON the device from (seed, batch counter), one
    float4 kernel mixes images and soft targets with the previous batch; nothing comes back to the host."""

    def __init__(self, seed=0):
        self.seed = int(seed)
        self.counter = 0
        self.key = None

    def reseed(self, random_seed, rank):
        """one stream per (run seed, rank): the reference draws from per-process np.random / torch generators, so ranks must
        not share their coins, lambdas, boxes and permutations"""
        self.seed = (int(random_seed or 0) * 1000003 + int(rank) * 7919 + 12345) & 0x7FFFFFFF

    def params_tensor(self):
        return self.params

    def __call__(self, data, target, cutmix_alpha, mixup_alpha, prob, allow):
        from . import native

        L = native.lib()
        if data.dtype != torch.float32 or target.dtype != torch.float32:
            raise TypeError(f"Mixup / CutMix on the device take float32 images and float32 soft targets (got {data.dtype}, {target.dtype})")
        data, target = data.contiguous(), target.contiguous()
        N, C, H, W = data.shape
        key = (tuple(data.shape), tuple(target.shape), data.device)
        if key != self.key:  # first batch / stage change: the "previous" batch is the batch itself (permuted)
            self.prev = [data.clone(), torch.empty_like(data)]
            self.tprev = [target.clone(), torch.empty_like(target)]
            self.params = torch.zeros(L.mi355_mix_params_bytes(N), dtype=torch.uint8, device=data.device)
            self.cur = 0
            self.key = key
        out, tout = torch.empty_like(data), torch.empty_like(target)
        st = native.cur_stream()
        native.check(L.mi355_mix_sample(native.ptr(self.params), self.seed, self.counter, N, H, W, float(cutmix_alpha), float(mixup_alpha),
                                        float(prob), int(allow), st))
        i, o = self.cur, self.cur ^ 1
        native.check(L.mi355_mix_apply(native.ptr(data), native.ptr(out), native.ptr(self.prev[i]), native.ptr(self.prev[o]),
                                       native.ptr(target), native.ptr(tout), native.ptr(self.tprev[i]), native.ptr(self.tprev[o]),
                                       native.ptr(self.params), N, C, H, W, target.shape[1], st))
        self.cur = o
        self.counter += 1
        return out, tout


class Mixup(Callback):
    def __init__(self, alpha, num_classes=1000, prob=0.5, seed=None):
        super().__init__()
        self._seed_given = seed is not None
        seed = seed or 0
        self.alpha = float(alpha)
        self.tb = torch.distributions.Beta(alpha, alpha)
        self.num_classes = num_classes
        self.prob = prob
        self.prev_input = None
        self._dev = _DeviceMixer(seed)

    def on_begin(self):
        if not self._seed_given:  # an explicit seed= stays; otherwise (run seed, rank) -> one stream per process
            self._dev.reseed(getattr(self.state, "random_seed", 0), self.state.rank)

    def on_loader_begin(self):
        # the sampler's position is a function of (epoch, step): a resumed run continues the sequence instead of replaying it
        if self.state.is_train and self.state.epoch_size:
            self._dev.counter = int(self.state.epoch) << 32  # (epoch, step) kept apart: epoch_size changes between progressive-resize stages

    def _onehot(self, target):
        if target.dim() == 1:
            return torch.nn.functional.one_hot(target.long(), self.num_classes).float()
        return target.float()

    def on_batch_begin(self):
        if self.state.is_train:
            self.state.input = self.mixup(*self.state.input)

    @torch.no_grad()
    def mixup(self, data, target):
        target = self._onehot(target)
        if data.is_cuda:  # the hot path: HIP kernels; the torch code below serves CPU tensors (host-logic tests) only
            a = float(self.tb.concentration1)
            return self._dev(data, target, a, a, self.prob, allow=1)
        if self.prev_input is None or self.prev_input[0].shape != data.shape:
            self.prev_input = (data.clone(), target.clone())
        if np.random.rand() > self.prob:
            self.prev_input = (data.clone(), target.clone())
            return data, target
        prev_data, prev_target = self.prev_input
        self.prev_input = (data.clone(), target.clone())
        perm = torch.randperm(data.size(0), device=data.device)
        c = float(self.tb.sample())
        return c * data + (1 - c) * prev_data[perm], c * target + (1 - c) * prev_target[perm]


class Cutmix(Mixup):
    def on_batch_begin(self):
        if self.state.is_train:
            self.state.input = self.cutmix(*self.state.input)

    @torch.no_grad()
    def cutmix(self, data, target):
        target = self._onehot(target)
        if data.is_cuda:
            a = float(self.tb.concentration1)
            return self._dev(data, target, a, a, self.prob, allow=2)
        if self.prev_input is None or self.prev_input[0].shape != data.shape:
            self.prev_input = (data.clone(), target.clone())
        if np.random.rand() > self.prob:
            self.prev_input = (data.clone(), target.clone())
            return data, target
        prev_data, prev_target = self.prev_input
        self.prev_input = (data.clone(), target.clone())
        _, _, H, W = data.shape
        lam = float(self.tb.sample())
        lam = min(lam, 1 - lam)
        bh, bw = int(H * np.sqrt(lam)), int(W * np.sqrt(lam))
        cy, cx = np.random.randint(H), np.random.randint(W)
        y1, y2 = np.clip(cy - bh // 2, 0, H), np.clip(cy + bh // 2, 0, H)
        x1, x2 = np.clip(cx - bw // 2, 0, W), np.clip(cx + bw // 2, 0, W)
        perm = torch.randperm(data.size(0), device=data.device)
        data = data.clone()
        data[:, :, y1:y2, x1:x2] = prev_data[perm][:, :, y1:y2, x1:x2]
        lam_real = float((y2 - y1) * (x2 - x1)) / (H * W)
        return data, (1 - lam_real) * target + lam_real * prev_target[perm]


class CutmixMixup(Cutmix):
    """sota_imagenet/callbacks.py:232-247."""

    def __init__(self, cutmix_alpha, mixup_alpha, prob=0.5, num_classes=1000, seed=None):
        super().__init__(cutmix_alpha, num_classes, prob, seed)
        self.cutmix_tb = torch.distributions.Beta(cutmix_alpha, cutmix_alpha)
        self.mixup_tb = torch.distributions.Beta(mixup_alpha, mixup_alpha)

    def on_batch_begin(self):
        if not self.state.is_train:
            return
        data, target = self.state.input
        if data.is_cuda:  # the coin of callbacks.py:242 is drawn on the device too (mi355_mix_sample, allow = both)
            self.state.input = self._dev(data, self._onehot(target), float(self.cutmix_tb.concentration1),
                                         float(self.mixup_tb.concentration1), self.prob, allow=3)
            return
        if np.random.rand() > 0.5:
            self.tb = self.cutmix_tb
            self.state.input = self.cutmix(*self.state.input)
        else:
            self.tb = self.mixup_tb
            self.state.input = self.mixup(*self.state.input)
