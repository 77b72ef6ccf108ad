"""Deterministic, framework-independent synthetic tensors: counter-based 32-bit integer hash -> values.

Integer arithmetic only (exact on CPU and GPU alike), so the torch-CPU oracle and the MI355X path consume
bit-identical batches and initial weights without shipping them.  Replaces, for benchmarking and parity,
what the reference gets from DALI (sota_imagenet/dali_dataloader.py:113-123: NCHW float32 batch normalised
with mean 127.5 / std 51.0 (:27-29), one-hot float labels) and from its un-vendored weight initialiser
(train.py:70-71 -> pytorch_tools.utils.misc.initialize, exact law not in the reference tree: initial
weights are therefore an explicit INPUT of both paths, generated here).
"""
import math
import zlib

import torch

DATA_MEAN = 127.5  # sota_imagenet/dali_dataloader.py:27
DATA_STD = 51.0    # sota_imagenet/dali_dataloader.py:29
_M32 = 0xFFFFFFFF


def hash32(idx, key):
    """lowbias32 on (idx + key * golden) mod 2^32.  idx: int64 tensor (any device). Returns int64 in [0, 2^32)."""
    x = (idx + (int(key) * 0x9E3779B9 & _M32)) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32  # int64 wrap-around keeps the low 32 bits exact
    x = x ^ (x >> 16)
    return x


def synthetic_batch(batch_size, image_size, num_classes=1000, seed=0, stream=0, index=0, device="cpu"):
    """One batch of the DALI contract: (data fp32 NCHW [N,3,S,S] = (u8-127.5)/51, target fp32 one-hot [N,C])."""
    n = batch_size * 3 * image_size * image_size
    key = (seed * 1000 + stream) * 65537 + index * 2 + 1
    idx = torch.arange(n, dtype=torch.int64, device=device)
    u8 = (hash32(idx, key) & 0xFF).to(torch.float32)
    data = ((u8 - DATA_MEAN) / DATA_STD).view(batch_size, 3, image_size, image_size)
    lab = hash32(torch.arange(batch_size, dtype=torch.int64, device=device), key + 1) % num_classes
    target = torch.zeros(batch_size, num_classes, dtype=torch.float32, device=device)
    target.scatter_(1, lab.view(-1, 1), 1.0)
    return data, target


def uniform_tensor(shape, bound, key):
    """U(-bound, bound) fp32 tensor of logical `shape` (row-major counter), generated on the CPU."""
    n = 1
    for s in shape:
        n *= s
    h = hash32(torch.arange(n, dtype=torch.int64), key).to(torch.float64)
    u = (h + 0.5) / 4294967296.0
    return ((2.0 * u - 1.0) * bound).to(torch.float32).view(*shape)


def init_state_dict(named_shapes, seed=0, gamma=1.72):
    """Initial parameters/buffers for a torchvision-layout ResNet: {name: fp32 tensor in torch logical shape}.

    conv / linear weights: U(-b, b) with b = gain * sqrt(3 / fan_in), gain = gamma for convs (train.py:70-71 passes
    init_gamma = 1.72, arg_parser.py:133) and 1/sqrt(3) for the classifier (torch's Linear default);
    BN weight 1 / bias 0 / running_mean 0 / running_var 1; linear bias U(-1/sqrt(fan_in), +).
    The law is this repo's choice (the reference's initialiser is not in its tree); both paths load the result.
    """
    out = {}
    for name, shape in named_shapes:
        key = (zlib.crc32(name.encode()) + 7919 * (seed + 1)) & 0x7FFFFFFF  # depends on the NAME, not on list order
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = uniform_tensor(shape, gamma * math.sqrt(3.0 / fan_in), key)
        elif len(shape) == 2:
            out[name] = uniform_tensor(shape, 1.0 / math.sqrt(shape[1]), key)
        elif name.startswith("fc.") and name.endswith("bias"):
            out[name] = uniform_tensor(shape, 1.0 / math.sqrt(2048.0), key)
        elif name.endswith("weight"):
            out[name] = torch.ones(shape)
        else:
            out[name] = torch.zeros(shape)
    return out
