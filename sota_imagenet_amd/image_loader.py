"""Real-image loader with the reference loader's protocol: JPEG folders -> the DALI pipeline's tensors, resize on the MI355X.

Mirrors sota_imagenet/dali_dataloader.py — `DaliLoader` (:163-186: `batch_size`, `__len__ = ceil(size / batch)`, `__iter__`
yielding `(data NCHW fp32, one-hot fp32)` on the device, last partial batch dropped :175, shard = rank :47) over
`fn.readers.file(file_root=root/"train" | root/"val")` (:68, :143: one sub-directory per class, sorted names = label ids).
Split of the work (DALI's "mixed" decoder + GPU operators):
  host    `workers` threads (:169 num_threads) decode with Pillow; for training the random crop is chosen BEFORE pixels are
          touched (`image_random_crop` :69-76: area in [min_area, 1], aspect ratio in [0.75, 1.25], 100 attempts) and only
          that rectangle is kept; the crops of a batch — all of different sizes — are packed back to back into one pinned
          buffer with a descriptor table
  device  ONE launch of mi355_ingest_u8 (csrc/ingest.hip): triangular-filter resize to S x S (train :78) or
          resize-shorter + centre window (val :144-149), mirror coin (:113-116), normalise with 127.5 / 51 (:27-29), NCHW fp32
The optional augmentations of the train pipeline (:78-114; all `*_prob = 0` by default, arg_parser.py:33-50; `ctwist: true` of
the legacy recipes maps to color_twist_prob) are drawn per sample on the host and applied by the same launch(es): cubic-or-
triangular resize by coin (`random_interpolation`), gaussian blur (window 11, sigma ~ U[0.5, 1.1]), colour twist (contrast /
brightness ~ U[0.7, 1.3], hue ~ U[-20, 20] degrees, saturation ~ U[0.7, 1.3]), grey coin, random erasing (`re_count`
rectangles, anchor ~ U[0, 1], side ~ U[0.05, 0.25] of the image, filled with the mean).  TFRecords are not supported.
There is no GPU JPEG decoder in this image (no rocJPEG), so end-to-end real-data throughput is bound by the host decode
(~150 img/s per core); the synthetic loader remains the benchmark feed (BASELINE.json: data = synthetic)."""
import math
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from .fit_wrapper import env_rank, env_world_size

DATA_MEAN, DATA_STD = 127.5, 51.0
IMG_EXT = (".jpeg", ".jpg", ".png", ".bmp", ".ppm", ".webp")
CROP_DTYPE = np.dtype([("offset", "<u8"), ("h", "<i4"), ("w", "<i4"), ("rh", "<i4"), ("rw", "<i4"), ("oy", "<i4"), ("ox", "<i4"),
                       ("mirror", "<i4"), ("filter", "<i4")])  # = mi355_crop (include/mi355rn.h), 40 bytes
AUG_DTYPE = np.dtype([("color", "<f4", (12,)), ("blur_sigma", "<f4"), ("gray", "<i4"), ("nbox", "<i4"), ("pad", "<i4"),
                      ("box", "<i4", (4, 4))])  # = mi355_augment, 128 bytes
assert CROP_DTYPE.itemsize == 40 and AUG_DTYPE.itemsize == 128
IDENTITY_COLOR = np.eye(3, 4, dtype=np.float32).reshape(12)
_RGB2YIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]])
_YIQ2RGB = np.array([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]])


def twist_matrix(brightness, contrast, hue_deg, saturation):
    """fn.color_twist (:89-98) as one 3 x 4 matrix on 0..255 values: chroma rotated by `hue_deg` and scaled by `saturation` in
    YIQ, contrast about 128, then brightness — v' = b * (128 + c * (HS v - 128))."""
    h = math.radians(hue_deg)
    rot = np.array([[1.0, 0.0, 0.0], [0.0, saturation * math.cos(h), -saturation * math.sin(h)], [0.0, saturation * math.sin(h), saturation * math.cos(h)]])
    m = np.zeros((3, 4))
    m[:, :3] = brightness * contrast * (_YIQ2RGB @ rot @ _RGB2YIQ)
    m[:, 3] = brightness * (1.0 - contrast) * 128.0
    return m.astype(np.float32).reshape(12)


def list_image_folder(root):
    """[(path, label)] of `root/<class>/<image>`: classes and files in sorted order (fn.readers.file semantics)."""
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    if not classes:
        raise FileNotFoundError(f"{root}: no class directories")
    samples = []
    for idx, c in enumerate(classes):
        d = os.path.join(root, c)
        samples += [(os.path.join(d, f), idx) for f in sorted(os.listdir(d)) if f.lower().endswith(IMG_EXT)]
    return samples, classes


def random_crop_box(rng, W, H, min_area=0.08, ratio=(0.75, 1.25), attempts=100):
    """(left, top, w, h) of image_random_crop: area ~ U[min_area, 1] of the image, aspect ratio w/h log-uniform in `ratio`,
    first attempt that fits wins; after `attempts` failures the largest centred rectangle with a legal aspect ratio."""
    for _ in range(attempts):
        area = rng.uniform(min_area, 1.0) * W * H
        ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        w, h = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
        if 0 < w <= W and 0 < h <= H:
            return int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h
    ar = min(max(W / H, ratio[0]), ratio[1])
    w, h = (W, int(round(W / ar))) if W / H <= ar else (int(round(H * ar)), H)
    w, h = min(w, W), min(h, H)
    return (W - w) // 2, (H - h) // 2, w, h


def val_geometry(h, w, S, full_crop=False):
    """(rh, rw, oy, ox) of the val pipeline (:144-157): shorter side -> ceil((S * 1.14 + 8) // 16 * 16), centre S x S window"""
    crop = S if full_crop else math.ceil((S * 1.14 + 8) // 16 * 16)
    if h <= w:
        rh, rw = crop, max(crop, int(round(w * crop / h)))
    else:
        rh, rw = max(crop, int(round(h * crop / w))), crop
    return rh, rw, (rh - S) // 2, (rw - S) // 2


class ImageFolderLoader:
    def __init__(self, cfg, is_val=False, seed=0, device=None, prefetch=2):
        from PIL import Image  # noqa: F401  (fail at construction, not in a worker thread)

        self.cfg = cfg
        self.is_val = bool(is_val)
        self._bs = int(cfg["batch_size"])
        self.image_size = int(cfg["image_size"])
        self.num_classes = int(cfg.get("num_classes", 1000))
        self.workers = max(1, int(cfg.get("workers", 6)))
        self.min_area = float(cfg.get("min_area", 0.08))
        self.full_crop = bool(cfg.get("full_crop", False))
        self.blur_prob, self.gray_prob = float(cfg.get("blur_prob", 0) or 0), float(cfg.get("gray_prob", 0) or 0)
        self.color_twist_prob, self.re_prob = float(cfg.get("color_twist_prob", 0) or 0), float(cfg.get("re_prob", 0) or 0)
        self.re_count = int(cfg.get("re_count", 3))
        self.contrast_range = tuple(cfg.get("contrast_range", (0.7, 1.3)))
        self.brightness_range = tuple(cfg.get("brightness_range", (0.7, 1.3)))
        self.random_interpolation = bool(cfg.get("random_interpolation", False))
        self.augmenting = not is_val and (self.blur_prob > 0 or self.gray_prob > 0 or self.color_twist_prob > 0 or self.re_prob > 0)
        if self.re_prob > 0 and not 1 <= self.re_count <= 4:
            raise ValueError(f"loader.re_count = {self.re_count}: the ingest kernel erases 1..4 rectangles per image")
        if cfg.get("use_tfrecords"):
            raise NotImplementedError("use_tfrecords is not supported by the MI355X ingest (folders only)")
        root = os.path.join(str(cfg["root_data_dir"]), "val" if is_val else "train")
        self.samples, self.classes = list_image_folder(root)
        if len(self.classes) > self.num_classes:
            raise ValueError(f"{root}: {len(self.classes)} class directories but num_classes = {self.num_classes}")
        self.rank, self.world = env_rank(), env_world_size()
        self.seed = int(seed or 0)
        self.epoch = 0
        self.device = torch.device(device if device is not None else "cuda")
        self.prefetch = max(1, int(prefetch))
        self._size = int(math.ceil(len(self.samples) / self.world))  # this shard's share (reader's num_shards)

    @property
    def batch_size(self):
        return self._bs

    def __len__(self):
        return math.ceil(self._size / self._bs)

    def batches_per_epoch(self):
        """full batches this loader yields per epoch — the same number on every rank"""
        return (len(self.samples) // self.world) // self._bs

    # ---- host half ---------------------------------------------------------------------------------------------------
    def _shard_indices(self, epoch):
        n = len(self.samples)
        order = np.random.default_rng((self.seed, epoch)).permutation(n) if not self.is_val else np.arange(n)
        return order[self.rank::self.world]

    def _decode(self, job):
        from PIL import Image

        idx, sample_seed = job
        path, label = self.samples[idx]
        S = self.image_size
        with Image.open(path) as im:
            W, H = im.size
            if self.is_val:
                box, mirror, filt, aug = (0, 0, W, H), 0, 0, None
            else:
                rng = np.random.default_rng(sample_seed)
                box = random_crop_box(rng, W, H, self.min_area)
                mirror = int(rng.integers(0, 2))
                filt = int(rng.integers(0, 2)) if self.random_interpolation else 0
                aug = self._draw_augment(rng) if self.augmenting else None
            px = np.asarray(im.convert("RGB").crop((box[0], box[1], box[0] + box[2], box[1] + box[3])), dtype=np.uint8)
        h, w = px.shape[:2]
        geo = val_geometry(h, w, S, self.full_crop) if self.is_val else (S, S, 0, 0)
        return px, geo, mirror, label, filt, aug

    def _draw_augment(self, rng):
        """one mi355_augment record: the coins and uniform draws of dali_dataloader.py:85-114 for one image"""
        S = self.image_size
        a = np.zeros((), dtype=AUG_DTYPE)
        a["color"] = IDENTITY_COLOR
        if self.blur_prob > 0 and rng.random() < self.blur_prob:
            a["blur_sigma"] = rng.uniform(0.5, 1.1)
        if self.color_twist_prob > 0 and rng.random() < self.color_twist_prob:
            a["color"] = twist_matrix(rng.uniform(*self.brightness_range), rng.uniform(*self.contrast_range), rng.uniform(-20.0, 20.0), rng.uniform(0.7, 1.3))
        if self.gray_prob > 0 and rng.random() < self.gray_prob:
            a["gray"] = 1
        if self.re_prob > 0 and rng.random() < self.re_prob:
            a["nbox"] = self.re_count
            for k in range(self.re_count):
                y0, x0 = int(rng.uniform(0.0, 1.0) * S), int(rng.uniform(0.0, 1.0) * S)
                hh, ww = int(round(rng.uniform(0.05, 0.25) * S)), int(round(rng.uniform(0.05, 0.25) * S))
                a["box"][k] = (y0, x0, min(y0 + hh, S), min(x0 + ww, S))
        return a

    def host_batch(self, indices, epoch, batch_no, pool):
        """decode + pack one batch: (packed u8 ndarray, CROP_DTYPE table, int64 labels, AUG_DTYPE table or None)"""
        jobs = [(int(i), (self.seed, epoch, int(i))) for i in indices]
        out = list(pool.map(self._decode, jobs))
        table = np.zeros(len(out), dtype=CROP_DTYPE)
        sizes = [o[0].size for o in out]
        offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])  # 16-byte aligned starts
        packed = np.zeros(int(offs[-1]), dtype=np.uint8)
        labels = np.zeros(len(out), dtype=np.int64)
        augs = np.zeros(len(out), dtype=AUG_DTYPE) if self.augmenting else None
        for n, (px, geo, mirror, label, filt, aug) in enumerate(out):
            packed[offs[n]:offs[n] + px.size] = px.reshape(-1)
            table[n] = (offs[n], px.shape[0], px.shape[1], geo[0], geo[1], geo[2], geo[3], mirror, filt)
            labels[n] = label
            if augs is not None:
                augs[n] = aug
        return packed, table, labels, augs

    # ---- device half -------------------------------------------------------------------------------------------------
    def to_device(self, packed, table, labels, augs=None):
        """upload + ONE ingest launch (two when a sample is blurred) on the current stream ->
        (data NCHW fp32 [N,3,S,S], one-hot fp32 [N,num_classes])"""
        dev = self.device
        p = torch.from_numpy(packed).pin_memory().to(dev, non_blocking=True)
        t = torch.from_numpy(table.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
        if augs is None:
            data = ops.ingest_u8(p, table, t, self.image_size, DATA_MEAN, DATA_STD)
        else:
            a = torch.from_numpy(augs.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
            data = ops.ingest_u8(p, table, t, self.image_size, DATA_MEAN, DATA_STD, aug_host=augs, aug_dev=a)
        lab = torch.from_numpy(labels).to(dev, non_blocking=True)
        onehot = torch.zeros((len(labels), self.num_classes), dtype=torch.float32, device=dev).scatter_(1, lab[:, None], 1.0)
        return data, onehot

    def __iter__(self):
        epoch = self.epoch
        self.epoch += 1
        order = self._shard_indices(epoch)
        # LastBatchPolicy.DROP — from the size every rank has in common: shards differ by one sample when the set does not
        # divide by the world size, and a rank with one batch more than its peers would wait in a gradient all-reduce forever
        n_full = self.batches_per_epoch()
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def producer():
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    for b in range(n_full):
                        if stop.is_set():
                            return
                        q.put(self.host_batch(order[b * self._bs:(b + 1) * self._bs], epoch, b, pool))
                q.put(None)
            except BaseException as e:  # surface decode errors in the consumer instead of hanging it
                q.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield self.to_device(*item)
        finally:
            stop.set()
            while th.is_alive():  # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
