"""Real-image ingest on the GPU (csrc/ingest.hip through mi355_ingest_u8) against the numpy oracle (oracle/ingest_ref.py, pinned
to Pillow's BILINEAR resize by tests/test_image_loader_host.py).  Tolerance: 2e-5 absolute on the normalised values (range
+-2.5): the kernel sums the same taps with the same weights in fp32, the oracle in fp64."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import ingest_ref as I

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-5


def _pack(crops, descs):
    from sota_imagenet_amd.image_loader import CROP_DTYPE

    table = np.zeros(len(crops), dtype=CROP_DTYPE)
    off, parts = 0, []
    for n, (px, d) in enumerate(zip(crops, descs)):
        table[n] = (off, px.shape[0], px.shape[1], d[0], d[1], d[2], d[3], d[4], d[5] if len(d) > 5 else 0)
        parts.append(px.reshape(-1))
        pad = (-px.size) % 16
        parts.append(np.zeros(pad, dtype=np.uint8))
        off += px.size + pad
    return np.concatenate(parts), table


def _run(crops, descs, S, dev, augs=None):
    from sota_imagenet_amd import ops

    packed, table = _pack(crops, descs)
    p = torch.from_numpy(packed).to(dev)
    t = torch.from_numpy(table.view(np.uint8)).to(dev)
    a = None if augs is None else torch.from_numpy(augs.view(np.uint8)).to(dev)
    return ops.ingest_u8(p, table, t, S, aug_host=augs, aug_dev=a).cpu().numpy(), (p, table, t)


def test_ingest_cubic_filter_and_every_augmentation(dev):
    """the optional operators of the train pipeline (dali_dataloader.py:78-114) against the oracle, alone and stacked; a batch
    with a blurred sample takes the two-launch path, one without stays on the single launch — both must agree with the oracle"""
    from sota_imagenet_amd.image_loader import AUG_DTYPE, IDENTITY_COLOR, twist_matrix

    rng = np.random.default_rng(1)
    S = 48
    sizes = [(120, 160), (30, 30), (200, 90), (64, 64), (48, 48), (333, 500)]
    crops = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    descs = [(S, S, 0, 0, n & 1, (n >> 1) & 1) for n in range(len(sizes))]  # mirror and filter in every combination
    augs = np.zeros(len(sizes), dtype=AUG_DTYPE)
    augs["color"] = IDENTITY_COLOR
    augs[1]["color"] = twist_matrix(1.25, 0.75, -18.0, 1.3)
    augs[2]["gray"] = 1
    augs[3]["nbox"] = 3
    augs[3]["box"][:3] = [(0, 0, 10, 7), (40, 20, 48, 48), (5, 30, 5, 40)]  # corner, clipped at the border, empty
    augs[4]["color"] = twist_matrix(0.7, 1.3, 20.0, 0.7)
    augs[4]["gray"] = 1
    augs[4]["nbox"] = 1
    augs[4]["box"][0] = (10, 10, 20, 30)

    def as_dict(a):
        return dict(color=np.asarray(a["color"], dtype=np.float64), blur_sigma=float(a["blur_sigma"]), gray=int(a["gray"]),
                    boxes=[tuple(int(v) for v in a["box"][k]) for k in range(int(a["nbox"]))])

    for blur in (False, True):
        if blur:
            augs[0]["blur_sigma"] = 0.8
            augs[4]["blur_sigma"] = 1.1
            augs[5]["blur_sigma"] = 0.5
        got, _ = _run(crops, descs, S, dev, augs)
        for n, (px, d) in enumerate(zip(crops, descs)):
            ref = I.ingest_one(px, d[0], d[1], d[2], d[3], S, d[4], filt=d[5], aug=as_dict(augs[n]))
            assert np.abs(got[n] - ref).max() < 3e-5, (blur, n)  # (the colour matrix is stored in fp32)
    # filter alone, no augment table: plain entry point
    got, _ = _run(crops, descs, S, dev)
    for n in (2, 3, 5):
        assert np.abs(got[n] - I.ingest_one(crops[n], S, S, 0, 0, S, descs[n][4], filt=descs[n][5])).max() < TOL
    bad = augs.copy()
    bad[0]["nbox"] = 5
    with pytest.raises(RuntimeError, match="nbox"):
        _run(crops, descs, S, dev, bad)


def test_ingest_matches_oracle_train_and_val_geometries(dev):
    rng = np.random.default_rng(0)
    S = 64
    sizes = [(300, 400), (40, 33), (64, 64), (1, 1), (7, 300), (500, 375), (90, 90), (65, 63), (2, 2), (128, 31)]
    crops = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    descs = []
    for n, (h, w) in enumerate(sizes):
        if n % 2 == 0:  # train: S x S, mirror coin
            descs.append((S, S, 0, 0, n % 4 // 2))
        else:  # val: resize-shorter to 80, centred window; plus off-centre windows and mirror (general descriptor)
            rh, rw = (80, max(80, round(w * 80 / h))) if h <= w else (max(80, round(h * 80 / w)), 80)
            descs.append((rh, rw, (rh - S) // 2, min(rw - S, 3 * n), n % 3 == 0))
    got, _ = _run(crops, descs, S, dev)
    assert got.shape == (len(crops), 3, S, S) and got.dtype == np.float32
    for n, (px, d) in enumerate(zip(crops, descs)):
        ref = I.ingest_one(px, d[0], d[1], d[2], d[3], S, d[4])
        assert np.abs(got[n] - ref).max() < TOL, (n, sizes[n], d)
    # the bare 1x1 source is a constant image: every output equals its normalised pixel
    assert np.allclose(got[3], ((crops[3][0, 0].astype(np.float64) - 127.5) / 51.0)[:, None, None], atol=1e-6)


@pytest.mark.parametrize("S", [160, 224, 320])
def test_ingest_at_the_recipe_sizes(dev, S):
    """progressive-resize sizes of BASELINE.json configs[4], typical ImageNet source shapes, down- and up-scaling"""
    rng = np.random.default_rng(S)
    sizes = [(375, 500), (500, 333), (224, 224), (1200, 1600), (100, 120), (333, 500)]
    crops = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    descs = [(S, S, 0, 0, n & 1) for n in range(len(sizes))]
    got, _ = _run(crops, descs, S, dev)
    for n in (0, 3, 4):  # the oracle's dense matrices are slow at these sizes: three samples
        assert np.abs(got[n] - I.ingest_one(crops[n], S, S, 0, 0, S, n & 1)).max() < TOL, (S, n)
    assert np.isfinite(got).all() and np.abs(got).max() <= 2.5 + 1e-6


def test_ingest_rejects_bad_descriptors(dev):
    from sota_imagenet_amd import ops

    px = np.zeros((8, 8, 3), dtype=np.uint8)
    _, (p, table, t) = _run([px], [(16, 16, 0, 0, 0)], 16, dev)
    for field, value, msg in [("offset", 64, "past the packed buffer"), ("oy", 1, "leaves"), ("h", 0, "empty"), ("mirror", 2, "mirror"), ("rw", 15, "leaves"),
                              ("filter", 2, "filter")]:
        bad = table.copy()
        bad[field] = value
        with pytest.raises(RuntimeError, match=msg):
            ops.ingest_u8(p, bad, t, 16)
    with pytest.raises(ValueError, match="40-byte"):
        ops.ingest_u8(p, table, t[:-1], 16)


def _make_folder(root, split, n_classes, per_class, seed):
    rng = np.random.default_rng(seed)
    for c in range(n_classes):
        d = os.path.join(root, split, f"n{c:04d}")
        os.makedirs(d)
        for k in range(per_class):
            h, w = int(rng.integers(48, 120)), int(rng.integers(48, 120))
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, f"img_{k}.jpg"), quality=90)


def test_folder_loader_end_to_end(dev, tmp_path):
    """JPEG folder -> (data, one-hot) on the device with the DaliLoader contract; the pixels equal the oracle applied to the
    loader's own decoded crops (same seeds)."""
    from concurrent.futures import ThreadPoolExecutor

    from sota_imagenet_amd import image_loader as L

    _make_folder(str(tmp_path), "train", 3, 6, 0)
    _make_folder(str(tmp_path), "val", 3, 2, 1)
    cfg = dict(batch_size=4, image_size=64, num_classes=1000, workers=3, root_data_dir=str(tmp_path))
    ld = L.ImageFolderLoader(cfg, seed=3)
    batches = list(ld)
    assert len(batches) == 18 // 4 == len(ld) - 1 + (18 % 4 == 0)  # the partial batch is dropped, __len__ is the ceil
    for data, target in batches:
        assert data.is_cuda and data.shape == (4, 3, 64, 64) and data.dtype == torch.float32
        assert target.is_cuda and target.shape == (4, 1000) and target.dtype == torch.float32
        assert torch.equal(target.sum(1), torch.ones(4, device=target.device)) and int(target.argmax(1).max()) < 3
        assert data.abs().max().item() <= 2.5 + 1e-6
    order = ld._shard_indices(0)
    with ThreadPoolExecutor(2) as pool:
        packed, table, labels, _ = ld.host_batch(order[:4], 0, 0, pool)
    assert torch.equal(batches[0][1].argmax(1).cpu(), torch.from_numpy(labels))
    for n in range(4):
        t = table[n]
        o, h, w = int(t["offset"]), int(t["h"]), int(t["w"])
        ref = I.ingest_one(packed[o:o + h * w * 3].reshape(h, w, 3), 64, 64, 0, 0, 64, int(t["mirror"]))
        assert np.abs(batches[0][0][n].cpu().numpy() - ref).max() < TOL
    second = list(ld)  # next epoch: reshuffled, re-cropped
    assert len(second) == len(batches) and not torch.equal(second[0][0], batches[0][0])
    # the full augmentation recipe of the later experiment configs (hydra_exp/10.*: blur / grey / twist / erase / random interpolation)
    la = L.ImageFolderLoader(dict(cfg, blur_prob=0.5, gray_prob=0.3, color_twist_prob=0.6, re_prob=0.5, re_count=3, random_interpolation=True), seed=3)
    ab = list(la)
    assert len(ab) == len(batches) and all(torch.isfinite(d).all() and d.abs().max().item() <= 2.5 + 1e-5 for d, _ in ab)
    with ThreadPoolExecutor(2) as pool:
        packed, table, labels, augs = la.host_batch(la._shard_indices(0)[:4], 0, 0, pool)
    assert augs is not None and augs.shape == (4,)
    for n in range(4):
        t, a = table[n], augs[n]
        o, h, w = int(t["offset"]), int(t["h"]), int(t["w"])
        ad = dict(color=np.asarray(a["color"], dtype=np.float64), blur_sigma=float(a["blur_sigma"]), gray=int(a["gray"]),
                  boxes=[tuple(int(v) for v in a["box"][k]) for k in range(int(a["nbox"]))])
        ref = I.ingest_one(packed[o:o + h * w * 3].reshape(h, w, 3), 64, 64, 0, 0, 64, int(t["mirror"]), filt=int(t["filter"]), aug=ad)
        assert np.abs(ab[0][0][n].cpu().numpy() - ref).max() < 3e-5
    lv = L.ImageFolderLoader(dict(cfg, batch_size=3), is_val=True, seed=3)
    vb = list(lv)
    assert len(vb) == 2 and vb[0][0].shape == (3, 3, 64, 64)
    px = np.asarray(Image.open(lv.samples[0][0]).convert("RGB"))
    rh, rw, oy, ox = L.val_geometry(px.shape[0], px.shape[1], 64)
    assert np.abs(vb[0][0][0].cpu().numpy() - I.ingest_one(px, rh, rw, oy, ox, 64, 0)).max() < TOL
    assert list(vb[0][1].argmax(1).cpu()) == [0, 0, 1]


def test_train_py_on_an_image_folder(dev, tmp_path):
    """the reference's entry point (train.py) fed from JPEG folders through the GPU ingest: one tiny epoch + validation"""
    sys.path.insert(0, ROOT)
    import train

    data_root = tmp_path / "imagenet"
    _make_folder(str(data_root), "train", 4, 8, 2)
    _make_folder(str(data_root), "val", 4, 4, 3)
    logdir = os.path.relpath(str(tmp_path / "logs"), ROOT)
    val_loss, metrics = train.main(["+hydra_exp=test", f"log.dir={logdir}", "random_seed=0", "data.source=folder", f"loader.root_data_dir={data_root}",
                                    f"val_loader.root_data_dir={data_root}", "loader.batch_size=8", "val_loader.batch_size=8", "loader.workers=2",
                                    "val_loader.workers=2"])
    assert val_loss == val_loss and 0.0 <= metrics["Acc@1"].avg <= 100.0
