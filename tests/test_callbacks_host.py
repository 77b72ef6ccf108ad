"""Mixup / CutMix / CutmixMixup (reference: sota_imagenet/callbacks.py:232-247) on CPU tensors:
the callbacks only rewrite `state.input`, so they are checked without the GPU path."""
import numpy as np
import torch

from sota_imagenet_amd import fit_wrapper as fw
from sota_imagenet_amd.callbacks import Cutmix, CutmixMixup, Mixup
from sota_imagenet_amd.synth import synthetic_batch


def _state(clb, batch):
    st = fw.RunnerState()
    st.is_train = True
    st.input = batch
    clb.set_state(st)
    return st


def test_mixup_produces_convex_soft_targets():
    torch.manual_seed(0)
    np.random.seed(0)
    clb = Mixup(alpha=0.4, num_classes=1000, prob=1.0)
    b0, b1 = synthetic_batch(4, 32, index=0), synthetic_batch(4, 32, index=1)
    st = _state(clb, b0)
    clb.on_batch_begin()  # first batch mixes with itself (prev_input = current)
    st.input = b1
    clb.on_batch_begin()
    data, target = st.input
    assert data.shape == b1[0].shape and target.shape == (4, 1000)
    assert torch.allclose(target.sum(1), torch.ones(4), atol=1e-6) and (target >= 0).all()
    assert ((target > 0).sum(1) <= 2).all()  # mixtures of two one-hot rows
    assert not torch.equal(data, b1[0])


def test_cutmix_target_weight_equals_box_area():
    torch.manual_seed(1)
    np.random.seed(1)
    clb = Cutmix(alpha=1.0, num_classes=1000, prob=1.0)
    b0, b1 = synthetic_batch(4, 32, index=2), synthetic_batch(4, 32, index=3)
    st = _state(clb, b0)
    clb.on_batch_begin()
    st.input = b1
    clb.on_batch_begin()
    data, target = st.input
    changed = (data != b1[0]).any(1).float().mean(dim=(1, 2))  # fraction of pixels replaced, per sample
    assert torch.allclose(target.sum(1), torch.ones(4), atol=1e-6)
    # the pasted box is the same for every sample: its area is the weight of the pasted label (lam_real)
    two = (target > 0).sum(1) == 2
    area = changed.max().item()
    for i in torch.nonzero(two).flatten().tolist():
        vals = target[i][target[i] > 0]
        assert min(abs(vals[0].item() - area), abs(vals[1].item() - area)) < 0.05  # noise pixels can coincide
    assert (changed <= 0.5 + 1e-6).all()  # lam = min(lam, 1-lam): at most half the image is replaced


def test_cutmixmixup_only_in_training_and_accepts_index_targets():
    np.random.seed(2)
    torch.manual_seed(2)
    clb = CutmixMixup(cutmix_alpha=1.0, mixup_alpha=0.2, prob=1.0, num_classes=10)
    data = torch.randn(6, 3, 16, 16)
    labels = torch.arange(6) % 10
    st = _state(clb, (data, labels))
    st.is_train = False
    clb.on_batch_begin()
    assert st.input[1] is labels  # evaluation batches are left alone
    st.is_train = True
    for _ in range(4):
        st.input = (data, labels)
        clb.on_batch_begin()
        d, t = st.input
        assert t.shape == (6, 10) and torch.allclose(t.sum(1), torch.ones(6), atol=1e-6)


def test_device_sampler_stream_is_per_seed_rank_and_epoch():
    """the device-side sampler's stream: (run seed, rank) -> seed, (epoch, step) -> counter; explicit seed= stays."""
    a, b = CutmixMixup(1.0, 0.2), CutmixMixup(1.0, 0.2)
    sa, sb = _state(a, None), _state(b, None)
    sa.random_seed = sb.random_seed = 42
    sa.rank, sb.rank = 0, 1
    a.on_begin(), b.on_begin()
    assert a._dev.seed != b._dev.seed and a._dev.seed != 0  # ranks do not share their coins / boxes / permutations
    sb.rank = 0
    b.on_begin()
    assert a._dev.seed == b._dev.seed
    sa.epoch, sa.epoch_size = 3, 100
    a.on_loader_begin()
    assert a._dev.counter == 3 << 32  # a resumed run continues the sequence; epochs of different lengths cannot overlap
    sa.is_train = False
    a._dev.counter = 7
    a.on_loader_begin()
    assert a._dev.counter == 7
    c = Mixup(0.4, seed=5)
    _state(c, None).random_seed = 42
    c.on_begin()
    assert c._dev.seed == 5


def test_device_mixer_rejects_non_float32():
    import pytest

    from sota_imagenet_amd.callbacks import _DeviceMixer

    with pytest.raises(TypeError):
        _DeviceMixer(0)(torch.zeros(2, 3, 8, 8, dtype=torch.bfloat16), torch.zeros(2, 10), 1.0, 1.0, 0.5, 3)
