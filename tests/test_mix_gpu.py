"""Device-side Mixup / CutMix (csrc/mix.hip) against a torch-CPU restatement of sota_imagenet/callbacks.py:232-247 + the
pytorch_tools Cutmix / Mixup bases, fed with the SAME lambda / permutation / box the device sampled
(read back from the parameter block in the test only), and properties of the on-device sampler."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def read_params(buf, N):
    raw = buf.cpu().numpy().tobytes()
    mode, = np.frombuffer(raw, np.int32, 1, 0)
    lam, = np.frombuffer(raw, np.float32, 1, 4)
    y1, y2, x1, x2 = np.frombuffer(raw, np.int32, 4, 8)
    lam_real, = np.frombuffer(raw, np.float32, 1, 24)
    perm = np.frombuffer(raw, np.int32, N, 32).copy()
    return int(mode), float(lam), int(y1), int(y2), int(x1), int(x2), float(lam_real), perm


def cpu_mix(data, target, prev, tprev, P):
    """the bases' arithmetic with given decisions: Mixup c*x + (1-c)*prev[perm]; CutMix pastes the box, target weights
    by the real box area"""
    mode, lam, y1, y2, x1, x2, lam_real, perm = P
    perm = torch.from_numpy(perm.astype(np.int64))
    if mode == 0:
        return data.clone(), target.clone()
    if mode == 1:
        return lam * data + (1 - lam) * prev[perm], lam * target + (1 - lam) * tprev[perm]
    out = data.clone()
    out[:, :, y1:y2, x1:x2] = prev[perm][:, :, y1:y2, x1:x2]
    return out, (1 - lam_real) * target + lam_real * tprev[perm]


@pytest.mark.parametrize("allow", [1, 2, 3])
def test_mix_apply_matches_cpu_restatement(dev, allow):
    from sota_imagenet_amd.callbacks import _DeviceMixer

    g = torch.Generator().manual_seed(5)
    N, C, H, W, K = 6, 3, 32, 48, 1000
    mixer = _DeviceMixer(seed=123)
    prev = tprev = None
    modes = set()
    for step in range(12):
        data = torch.randn(N, C, H, W, generator=g)
        lab = torch.randint(0, K, (N,), generator=g)
        target = torch.nn.functional.one_hot(lab, K).float()
        if prev is None:
            prev, tprev = data, target  # first batch: mixed with itself, permuted
        out, tout = mixer(data.to(dev), target.to(dev), 1.0, 0.2, 0.9, allow)
        P = read_params(mixer.params_tensor(), N)
        modes.add(P[0])
        assert sorted(P[7].tolist()) == list(range(N)), "perm is a permutation"
        ref, tref = cpu_mix(data, target, prev, tprev, P)
        assert (out.cpu() - ref).abs().max().item() <= 1e-6 * max(ref.abs().max().item(), 1.0), (step, P[:7])
        assert (tout.cpu() - tref).abs().max().item() <= 1e-6, (step, P[:7])
        assert abs(tout.sum(1).cpu() - 1.0).max().item() < 1e-5  # soft targets stay distributions
        prev, tprev = data, target  # the UNMIXED batch is what the next step mixes with
    assert modes - {0} <= ({1} if allow == 1 else {2} if allow == 2 else {1, 2}) and len(modes - {0}) >= 1


def test_mix_sampler_properties(dev):
    from sota_imagenet_amd import native

    L = native.lib()
    N, H, W = 64, 224, 224
    buf = torch.zeros(L.mi355_mix_params_bytes(N), dtype=torch.uint8, device=dev)

    def draw(seed, counter, ca=1.0, ma=0.2, prob=1.0, allow=3):
        native.check(L.mi355_mix_sample(native.ptr(buf), seed, counter, N, H, W, ca, ma, prob, allow, native.cur_stream()))
        return read_params(buf, N)

    a, b = draw(7, 3), draw(7, 3)
    assert a[:7] == b[:7] and (a[7] == b[7]).all(), "same (seed, counter) -> same decisions"
    c = draw(7, 4)
    assert not (a[7] == c[7]).all()
    S = [draw(11, k) for k in range(400)]
    modes = np.array([s[0] for s in S])
    assert 0.35 < (modes == 2).mean() < 0.65 and (modes == 0).sum() == 0  # the coin of callbacks.py:242, prob = 1
    lam_cut = np.array([s[1] for s in S if s[0] == 2])   # Beta(1, 1) = uniform
    lam_mix = np.array([s[1] for s in S if s[0] == 1])   # Beta(0.2, 0.2): U-shaped, mass near 0 and 1
    assert ((lam_cut > 0) & (lam_cut < 1)).all() and abs(lam_cut.mean() - 0.5) < 0.08 and abs(lam_cut.var() - 1 / 12) < 0.03
    assert ((lam_mix >= 0) & (lam_mix <= 1)).all() and abs(lam_mix.mean() - 0.5) < 0.1
    assert abs(lam_mix.var() - 0.25 / (2 * 0.2 + 1)) < 0.05  # var of Beta(a, a) = 1 / (4 (2a + 1))
    for s in S:
        if s[0] == 2:
            _, lam, y1, y2, x1, x2, lr, _ = s
            assert 0 <= y1 <= y2 <= H and 0 <= x1 <= x2 <= W
            assert abs(lr - (y2 - y1) * (x2 - x1) / (H * W)) < 1e-6
            assert (y2 - y1) <= int(H * np.sqrt(min(lam, 1 - lam))) + 1
    off = [draw(13, k, prob=0.3)[0] for k in range(300)]
    assert 0.18 < np.mean(np.array(off) != 0) < 0.42  # applied with probability `prob`
