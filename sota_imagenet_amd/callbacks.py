"""Input-side callbacks next to the hot path: Mixup / CutMix / CutmixMixup on device, producing the soft targets the
native cross entropy consumes.

Re-states sota_imagenet/callbacks.py:232-247 (`CutmixMixup`: coin flip between `self.cutmix(*input)` and
`self.mixup(*input)` with Beta(alpha, alpha) samplers) and the un-vendored pt_clb.Cutmix / pt_clb.Mixup bases as
SURVEY.md Appendix C records them (mix with the PREVIOUS batch, permuted; CutMix target weight = real box area).
"""
import numpy as np
import torch

from .fit_wrapper import Callback


class Mixup(Callback):
    def __init__(self, alpha, num_classes=1000, prob=0.5):
        super().__init__()
        self.tb = torch.distributions.Beta(alpha, alpha)
        self.num_classes = num_classes
        self.prob = prob
        self.prev_input = None

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

    def __init__(self, cutmix_alpha, mixup_alpha, prob=0.5, num_classes=1000):
        super().__init__(cutmix_alpha, num_classes, prob)
        self.cutmix_tb = torch.distributions.Beta(cutmix_alpha, cutmix_alpha)
        self.mixup_tb = torch.distributions.Beta(mixup_alpha, mixup_alpha)

    def on_batch_begin(self):
        if not self.state.is_train:
            return
        if np.random.rand() > 0.5:
            self.tb = self.cutmix_tb
            self.state.input = self.cutmix(*self.state.input)
        else:
            self.tb = self.mixup_tb
            self.state.input = self.mixup(*self.state.input)
