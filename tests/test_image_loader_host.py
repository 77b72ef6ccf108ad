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


@pytest.mark.parametrize("shape", [(300, 400, 224, 224), (100, 80, 224, 224), (37, 53, 64, 64), (224, 224, 224, 224), (700, 500, 128, 128)])
def test_oracle_cubic_matches_pillow_bicubic(shape):
    """the coin of `random_interpolation` (:79-83): the a = -0.5 cubic against Pillow's BICUBIC (mid-range pixels: Pillow clips
    its 8-bit intermediate, the oracle only the result)"""
    h, w, rh, rw = shape
    img = np.random.default_rng(h + w).integers(64, 192, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((rw, rh), Image.BICUBIC)).astype(np.float64)
    assert np.abs(I.resize(img, rh, rw, 1) - ref).max() <= 1.1, shape


def test_oracle_blur_colour_grey_erase():
    import colorsys

    import scipy.ndimage as ndi

    rng = np.random.default_rng(5)
    win = rng.uniform(0, 255, (40, 40, 3))
    for sigma in (0.5, 0.8, 1.1):
        d = np.arange(-5, 6)
        w = np.exp(-0.5 * (d / sigma) ** 2)
        w /= w.sum()
        ref = ndi.correlate1d(ndi.correlate1d(win, w, axis=0, mode="mirror"), w, axis=1, mode="mirror")  # scipy "mirror" = reflect-101
        assert np.abs(I.gaussian_blur(win, sigma) - ref).max() < 1e-9
    # colour: the YIQ transform is the one of the standard library; neutral parameters are (nearly) the identity; saturation 0
    # leaves luma only; contrast 0 gives uniform 128 * brightness; product and oracle agree
    for rgb in [(0.2, 0.5, 0.9), (1.0, 0.0, 0.3)]:
        assert np.allclose(I.RGB2YIQ @ np.array(rgb), colorsys.rgb_to_yiq(*rgb), atol=6e-3)
    assert np.abs(I.twist_matrix()[:, :3] - np.eye(3)).max() < 2e-3 and not I.twist_matrix()[:, 3].any()
    m0 = I.twist_matrix(saturation=0.0)
    assert np.allclose(m0[:, :3], np.tile(I.LUMA, (3, 1)), atol=1e-9)
    mc = I.twist_matrix(brightness=1.2, contrast=0.0)
    assert np.allclose(mc[:, :3], 0.0) and np.allclose(mc[:, 3], 1.2 * 128.0)
    assert np.allclose(L.twist_matrix(1.1, 0.8, 15.0, 1.2).reshape(3, 4), I.twist_matrix(1.1, 0.8, 15.0, 1.2), atol=1e-5)
    out = I.augment(win, dict(color=I.twist_matrix(1.3, 1.3, 10.0, 1.3), gray=1, boxes=[(2, 3, 10, 12), (30, 30, 40, 40)]))
    assert out.min() >= 0.0 and out.max() <= 255.0 and np.allclose(out[..., 0], out[..., 1])
    assert np.all(out[2:10, 3:12] == I.DATA_MEAN) and np.all(out[30:, 30:] == I.DATA_MEAN) and not np.all(out[0, 0] == I.DATA_MEAN)


def test_augment_draws_follow_the_recipe(tmp_path):
    _make_folder(str(tmp_path), "train", n_classes=1, per_class=1)
    cfg = dict(batch_size=1, image_size=64, num_classes=10, workers=1, root_data_dir=str(tmp_path), blur_prob=0.5, gray_prob=0.3, color_twist_prob=0.6,
               re_prob=0.4, re_count=3, random_interpolation=True)
    ld = L.ImageFolderLoader(cfg, device="cpu")
    assert ld.augmenting
    recs = [ld._draw_augment(np.random.default_rng(k)) for k in range(400)]
    blur = np.array([float(r["blur_sigma"]) for r in recs])
    assert 0.4 < (blur > 0).mean() < 0.6 and blur[blur > 0].min() >= 0.5 and blur.max() <= 1.1
    assert 0.2 < np.mean([int(r["gray"]) for r in recs]) < 0.4
    tw = np.array([not np.array_equal(r["color"], L.IDENTITY_COLOR) for r in recs])
    assert 0.5 < tw.mean() < 0.7
    er = [r for r in recs if r["nbox"]]
    assert 0.3 < len(er) / 400 < 0.5 and all(int(r["nbox"]) == 3 for r in er)
    for r in er:
        for (y0, x0, y1, x1) in r["box"][:3]:
            assert 0 <= y0 <= y1 <= 64 and 0 <= x0 <= x1 <= 64 and y1 - y0 <= 16 and x1 - x0 <= 16  # sides <= 0.25 * S
    assert L.ImageFolderLoader(dict(cfg, blur_prob=0, gray_prob=0, color_twist_prob=0, re_prob=0), device="cpu").augmenting is False
    assert L.ImageFolderLoader(cfg, is_val=False, device="cpu").random_interpolation


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
        packed, table, labels, augs = ld.host_batch(order[:4], 0, 0, pool)
        packed2, table2, _, _ = ld.host_batch(order[:4], 0, 0, pool)
    assert augs is None and not table["filter"].any()  # every optional augmentation is off by default
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
    # 15 images over 2 ranks: shards of 8 and 7 — every rank must yield the SAME number of batches (a rank with one batch more
    # would sit in a gradient all-reduce without a peer); checked at every batch size around the boundary
    for bs in (1, 2, 3, 4, 7, 8):
        counts = []
        for r in (0, 1):
            monkeypatch.setenv("RANK", str(r))
            lr = L.ImageFolderLoader(dict(cfg, batch_size=bs), is_val=False, seed=7, device="cpu")
            counts.append(lr.batches_per_epoch())
            assert counts[-1] * bs <= len(lr._shard_indices(0))
        assert counts[0] == counts[1] == (15 // 2) // bs, (bs, counts)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    lv = L.ImageFolderLoader(dict(cfg, batch_size=3), is_val=True, seed=7, device="cpu")
    assert np.array_equal(lv._shard_indices(0), np.arange(6)) and len(lv) == 2
    with ThreadPoolExecutor(2) as pool:
        packed, table, labels, _ = lv.host_batch(np.arange(3), 0, 0, pool)
    for n in range(3):
        full = np.asarray(Image.open(lv.samples[n][0]).convert("RGB"))
        t = table[n]
        assert (t["h"], t["w"]) == full.shape[:2] and t["mirror"] == 0
        assert (t["rh"], t["rw"], t["oy"], t["ox"]) == L.val_geometry(full.shape[0], full.shape[1], 32)


def test_unsupported_augmentations_raise_and_source_selection(tmp_path):
    from sota_imagenet_amd import data

    _make_folder(str(tmp_path), "train", n_classes=2, per_class=2)
    cfg = dict(batch_size=2, image_size=32, num_classes=10, workers=1, root_data_dir=str(tmp_path))
    with pytest.raises(NotImplementedError, match="use_tfrecords"):
        L.ImageFolderLoader(dict(cfg, use_tfrecords=True), device="cpu")
    with pytest.raises(ValueError, match="re_count"):
        L.ImageFolderLoader(dict(cfg, re_prob=0.5, re_count=7), device="cpu")
    with pytest.raises(ValueError, match="class directories"):
        L.ImageFolderLoader(dict(cfg, num_classes=1), device="cpu")
    assert isinstance(data.make_loader(cfg, 100, 0, "cpu", 2, False, "auto"), L.ImageFolderLoader)
    assert isinstance(data.make_loader(cfg, 100, 0, "cpu", 2, True, "auto"), data.SyntheticLoader)  # no val/ directory
    assert isinstance(data.make_loader(cfg, 100, 0, "cpu", 2, False, "synthetic"), data.SyntheticLoader)
    assert isinstance(data.make_loader(dict(cfg, root_data_dir=""), 100, 0, "cpu", 2, False), data.SyntheticLoader)
    with pytest.raises(FileNotFoundError):
        data.make_loader(dict(cfg, root_data_dir=str(tmp_path / "nope")), 100, 0, "cpu", 2, False, "folder")
    assert data.DaliDataManager is data.SyntheticDataManager
