"""Host half of the real-image ingest (sota_imagenet_amd/image_loader.py) and the pin of its oracle (oracle/ingest_ref.py):
  * the oracle's triangular resize against Pillow's own Image.resize(BILINEAR) — an independent implementation of the same
    filter law — within 1 LSB (Pillow keeps an 8-bit intermediate between its two passes);
  * crop-box / val-geometry laws of sota_imagenet/dali_dataloader.py:69-76,144-149;
  * the loader protocol (:163-186): folder listing, rank shard, drop-last, packed bytes == the Pillow crop, labels."""
import math
import os

import numpy as np
import pytest
from PIL import Image

from oracle import ingest_ref as I
from sota_imagenet_amd import image_loader as L


@pytest.mark.parametrize("shape", [(300, 400, 224, 224), (100, 80, 224, 224), (500, 375, 160, 160), (37, 53, 64, 64), (224, 224, 224, 224),
                                   (700, 500, 128, 128), (64, 64, 65, 63), (9, 7, 32, 32), (1, 1, 8, 8), (640, 480, 320, 320)])
def test_oracle_resize_matches_pillow_bilinear(shape):
    h, w, rh, rw = shape
    rng = np.random.default_rng(h * 1000 + w)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
    for img in (noise, smooth):
        ref = np.asarray(Image.fromarray(img).resize((rw, rh), Image.BILINEAR)).astype(np.float64)
        got = I.resize(img, rh, rw)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1.0 + 1e-3, shape  # + Pillow's 22-bit fixed-point coefficients
    # constant images stay constant (weights are normalised at the borders)
    const = np.full((h, w, 3), 137, dtype=np.uint8)
    assert np.abs(I.resize(const, rh, rw) - 137.0).max() < 1e-9


def test_oracle_window_and_mirror_are_slices_of_the_full_resize():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    full = (I.resize(img, 72, 104) - I.DATA_MEAN) / I.DATA_STD
    got = I.ingest_one(img, 72, 104, 4, 20, 64, 0)
    assert np.allclose(got, full[4:68, 20:84].transpose(2, 0, 1), atol=1e-6)
    got_m = I.ingest_one(img, 72, 104, 4, 20, 64, 1)
    assert np.allclose(got_m, full[4:68, 20:84][:, ::-1].transpose(2, 0, 1), atol=1e-6)
    assert got.dtype == np.float32 and abs(I.DATA_MEAN - 0.5 * 255) < 1e-12 and abs(I.DATA_STD - 0.2 * 255) < 1e-12


def test_random_crop_box_law():
    rng = np.random.default_rng(0)
    for (W, H) in [(500, 375), (375, 500), (64, 64), (1000, 50), (3, 2)]:
        for _ in range(200):
            x, y, w, h = L.random_crop_box(rng, W, H, 0.08)
            assert 0 <= x and 0 <= y and x + w <= W and y + h <= H and w > 0 and h > 0
            if min(W, H) >= 32 and 0.7 < W / H < 1.4:  # an attempt fits: bounds of image_random_crop (up to integer rounding)
                assert 0.06 * W * H <= w * h <= W * H
                assert 0.70 <= w / h <= 1.33
    # the fallback (no attempt fits a 1000 x 50 strip at large areas) stays inside the image and legal
    x, y, w, h = L.random_crop_box(np.random.default_rng(1), 1000, 50, 0.9)
    assert (w, h) == (62, 50) or (0.75 <= w / h <= 1.25 + 1e-9)
    a = L.random_crop_box(np.random.default_rng(5), 500, 375)
    assert a == L.random_crop_box(np.random.default_rng(5), 500, 375)


def test_val_geometry():
    assert math.ceil((224 * 1.14 + 8) // 16 * 16) == 256  # the reference's formula (:147) at 224 px
    assert L.val_geometry(375, 500, 224) == (256, 341, 16, 58)
    assert L.val_geometry(500, 375, 224) == (341, 256, 58, 16)
    assert L.val_geometry(300, 300, 224, full_crop=True) == (224, 224, 0, 0)
    assert L.val_geometry(100, 100, 288) == (336, 336, 24, 24)
    assert I.val_geometry(375, 500, 224) == L.val_geometry(375, 500, 224)


def _make_folder(root, split, n_classes=3, per_class=5, seed=0):
    rng = np.random.default_rng(seed)
    for c in range(n_classes):
        d = os.path.join(root, split, f"n{c:04d}")
        os.makedirs(d)
        for k in range(per_class):
            h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, f"img_{k}.png"))


def test_loader_protocol_and_packing(tmp_path, monkeypatch):
    from concurrent.futures import ThreadPoolExecutor

    _make_folder(str(tmp_path), "train")
    _make_folder(str(tmp_path), "val", per_class=2, seed=1)
    cfg = dict(batch_size=4, image_size=32, num_classes=10, workers=2, root_data_dir=str(tmp_path), min_area=0.08)
    ld = L.ImageFolderLoader(cfg, is_val=False, seed=7, device="cpu")
    assert ld.batch_size == 4 and len(ld.samples) == 15 and len(ld) == math.ceil(15 / 4) and ld.classes == ["n0000", "n0001", "n0002"]
    order = ld._shard_indices(0)
    assert sorted(order.tolist()) == list(range(15)) and not np.array_equal(order, np.arange(15))
    assert not np.array_equal(order, ld._shard_indices(1))  # reshuffled every epoch (random_shuffle=True :56)
    with ThreadPoolExecutor(2) as pool:
        packed, table, labels = ld.host_batch(order[:4], 0, 0, pool)
        packed2, table2, _ = ld.host_batch(order[:4], 0, 0, pool)
    assert np.array_equal(packed, packed2) and np.array_equal(table, table2)  # crops are a function of (seed, epoch, sample)
    assert table.dtype.itemsize == 40 and table.shape == (4,)
    for n, idx in enumerate(order[:4]):
        path, label = ld.samples[idx]
        assert labels[n] == label == int(os.path.basename(os.path.dirname(path))[1:])
        t = table[n]
        assert t["offset"] % 16 == 0 and (t["rh"], t["rw"], t["oy"], t["ox"]) == (32, 32, 0, 0) and t["mirror"] in (0, 1)
        full = np.asarray(Image.open(path).convert("RGB"))
        o, hh, ww = int(t["offset"]), int(t["h"]), int(t["w"])
        crop = packed[o:o + hh * ww * 3].reshape(hh, ww, 3)
        rng = np.random.default_rng((7, 0, int(idx)))
        x, y, w, h = L.random_crop_box(rng, full.shape[1], full.shape[0], 0.08)
        assert (hh, ww) == (h, w) and np.array_equal(crop, full[y:y + h, x:x + w])
    # rank shards are disjoint and cover the set; validation is unshuffled and uses the resize-shorter geometry
    monkeypatch.setenv("WORLD_SIZE", "2")
    shards = []
    for r in (0, 1):
        monkeypatch.setenv("RANK", str(r))
        monkeypatch.setenv("LOCAL_RANK", str(r))
        lr = L.ImageFolderLoader(cfg, is_val=False, seed=7, device="cpu")
        shards.append(set(lr._shard_indices(0).tolist()))
        assert len(lr) == math.ceil(math.ceil(15 / 2) / 4)
    assert shards[0] | shards[1] == set(range(15)) and not (shards[0] & shards[1])
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    lv = L.ImageFolderLoader(dict(cfg, batch_size=3), is_val=True, seed=7, device="cpu")
    assert np.array_equal(lv._shard_indices(0), np.arange(6)) and len(lv) == 2
    with ThreadPoolExecutor(2) as pool:
        packed, table, labels = lv.host_batch(np.arange(3), 0, 0, pool)
    for n in range(3):
        full = np.asarray(Image.open(lv.samples[n][0]).convert("RGB"))
        t = table[n]
        assert (t["h"], t["w"]) == full.shape[:2] and t["mirror"] == 0
        assert (t["rh"], t["rw"], t["oy"], t["ox"]) == L.val_geometry(full.shape[0], full.shape[1], 32)


def test_unsupported_augmentations_raise_and_source_selection(tmp_path):
    from sota_imagenet_amd import data

    _make_folder(str(tmp_path), "train", n_classes=2, per_class=2)
    cfg = dict(batch_size=2, image_size=32, num_classes=10, workers=1, root_data_dir=str(tmp_path))
    with pytest.raises(NotImplementedError, match="blur_prob"):
        L.ImageFolderLoader(dict(cfg, blur_prob=0.3), device="cpu")
    with pytest.raises(ValueError, match="class directories"):
        L.ImageFolderLoader(dict(cfg, num_classes=1), device="cpu")
    assert isinstance(data.make_loader(cfg, 100, 0, "cpu", 2, False, "auto"), L.ImageFolderLoader)
    assert isinstance(data.make_loader(cfg, 100, 0, "cpu", 2, True, "auto"), data.SyntheticLoader)  # no val/ directory
    assert isinstance(data.make_loader(cfg, 100, 0, "cpu", 2, False, "synthetic"), data.SyntheticLoader)
    assert isinstance(data.make_loader(dict(cfg, root_data_dir=""), 100, 0, "cpu", 2, False), data.SyntheticLoader)
    with pytest.raises(FileNotFoundError):
        data.make_loader(dict(cfg, root_data_dir=str(tmp_path / "nope")), 100, 0, "cpu", 2, False, "folder")
    assert data.DaliDataManager is data.SyntheticDataManager
