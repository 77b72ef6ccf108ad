"""ORACLE (test infrastructure — never imported by the product path).

Plain-`torch.nn` CPU restatement of the model / loss / optimizer / LR-schedule the reference wires together:
  model     torchvision-layout ResNet-50 v1.5 ("default torchvision version of Resnet50", README.md:42;
            `_target_: pytorch_tools.models.resnet50`, configs/hydra_exp/1.r50_baseline.yaml:22-23) — the
            package is not vendored, so the architecture is restated from its published definition
            (SURVEY.md §A.3 shape table; 25 557 032 parameters = "25.56M", 1.r50_baseline.yaml:11)
  loss      label-smoothed CE on float targets (sota_imagenet/arg_parser.py:140-142; smoothing 0.1,
            1.r50_baseline.yaml:34-35)
  optimizer torch.optim.SGD momentum 0.9 wd 3e-5 (arg_parser.py:136-138; 1.r50_baseline.yaml:29-31)
  step      autocast-free fp32 restatement of the Runner inner step the reference re-enacts in
            sota_imagenet/callbacks.py:314-317 (criterion(model(data), target) -> backward -> step)
PARITY UNPINNED by the reference (it has no tests or golden vectors for this path — SURVEY.md §4, §8c):
this file is pinned only by the parameter count above and by torch's CPU kernels; fixtures generated from it
live in tests/golden/ (tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return F.relu(out + idt)


class ResNet50Ref(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.fc = nn.Linear(2048, num_classes)

    def _make_layer(self, planes, blocks, stride):
        ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(x)


def patch_bn_mom(model, momentum):
    """train.py:76 — pt.utils.misc.patch_bn_mom(model, cfg.bn_momentum) (0.1, arg_parser.py:132)."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = momentum


def smooth_ce(logits, target, smoothing=0.1):
    logp = F.log_softmax(logits.float(), dim=1)
    return ((1.0 - smoothing) * -(logp * target).sum(1) + smoothing * -logp.mean(1)).mean()


def phase_lr(lr_stages, epoch, step, epoch_size):
    """pt_clb.PhasesScheduler restated (train.py:117-131): lr_stages =
    [{ep:(start,end), lr:(a,b)|scalar, mode:"linear"|"cos"}]; the phase is the last one whose start <= epoch."""
    phase = None
    for p in lr_stages:
        if p["ep"][0] <= epoch:
            phase = p
    start, end = phase["ep"]
    lr = phase["lr"]
    a, b = (lr, lr) if not isinstance(lr, (list, tuple)) else (lr[0], lr[-1])
    pct = ((epoch - start) * epoch_size + step) / float((end - start) * epoch_size)
    if phase.get("mode", "linear") == "cos":
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    return a + (b - a) * pct


def make_reference(state_dict, num_classes=1000, bn_momentum=0.1):
    m = ResNet50Ref(num_classes)
    missing = m.load_state_dict(state_dict, strict=True)
    patch_bn_mom(m, bn_momentum)
    return m


def train_steps(model, batches, lrs, momentum=0.9, weight_decay=3e-5, smoothing=0.1):
    """Runs len(batches) fp32 SGD steps; returns ([loss per step], logits of the first step)."""
    opt = torch.optim.SGD(model.parameters(), lr=0.0, momentum=momentum, weight_decay=weight_decay)
    model.train()
    losses, first_logits = [], None
    for (data, target), lr in zip(batches, lrs):
        for g in opt.param_groups:
            g["lr"] = lr
        out = model(data)
        if first_logits is None:
            first_logits = out.detach().clone()
        loss = smooth_ce(out, target, smoothing)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses, first_logits
