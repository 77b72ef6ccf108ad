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
Augmentations that are off in every r50 recipe (blur, colour twist, grey, random erasing: :80-109, all `*_prob = 0` defaults,
arg_parser.py:33-50) are not implemented: a non-zero probability raises instead of being ignored.
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
                       ("mirror", "<i4"), ("pad", "<i4")])  # = mi355_crop (include/mi355rn.h), 40 bytes
assert CROP_DTYPE.itemsize == 40


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
        for k in ("blur_prob", "gray_prob", "color_twist_prob", "re_prob"):
            if float(cfg.get(k, 0) or 0) > 0:
                raise NotImplementedError(f"loader.{k} > 0: this augmentation is not part of the MI355X ingest (defaults are 0)")
        if cfg.get("random_interpolation") or cfg.get("use_tfrecords"):
            raise NotImplementedError("random_interpolation / use_tfrecords are not supported by the MI355X ingest")
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
                box, mirror = (0, 0, W, H), 0
            else:
                rng = np.random.default_rng(sample_seed)
                box = random_crop_box(rng, W, H, self.min_area)
                mirror = int(rng.integers(0, 2))
            px = np.asarray(im.convert("RGB").crop((box[0], box[1], box[0] + box[2], box[1] + box[3])), dtype=np.uint8)
        h, w = px.shape[:2]
        geo = val_geometry(h, w, S, self.full_crop) if self.is_val else (S, S, 0, 0)
        return px, geo, mirror, label

    def host_batch(self, indices, epoch, batch_no, pool):
        """decode + pack one batch: (packed u8 ndarray, CROP_DTYPE table, int64 labels)"""
        jobs = [(int(i), (self.seed, epoch, int(i))) for i in indices]
        out = list(pool.map(self._decode, jobs))
        table = np.zeros(len(out), dtype=CROP_DTYPE)
        sizes = [o[0].size for o in out]
        offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])  # 16-byte aligned starts
        packed = np.zeros(int(offs[-1]), dtype=np.uint8)
        labels = np.zeros(len(out), dtype=np.int64)
        for n, (px, geo, mirror, label) in enumerate(out):
            packed[offs[n]:offs[n] + px.size] = px.reshape(-1)
            table[n] = (offs[n], px.shape[0], px.shape[1], geo[0], geo[1], geo[2], geo[3], mirror, 0)
            labels[n] = label
        return packed, table, labels

    # ---- device half -------------------------------------------------------------------------------------------------
    def to_device(self, packed, table, labels):
        """upload + ONE ingest launch on the current stream -> (data NCHW fp32 [N,3,S,S], one-hot fp32 [N,num_classes])"""
        dev = self.device
        p = torch.from_numpy(packed).pin_memory().to(dev, non_blocking=True)
        t = torch.from_numpy(table.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
        data = ops.ingest_u8(p, table, t, self.image_size, DATA_MEAN, DATA_STD)
        lab = torch.from_numpy(labels).to(dev, non_blocking=True)
        onehot = torch.zeros((len(labels), self.num_classes), dtype=torch.float32, device=dev).scatter_(1, lab[:, None], 1.0)
        return data, onehot

    def __iter__(self):
        epoch = self.epoch
        self.epoch += 1
        order = self._shard_indices(epoch)
        n_full = len(order) // self._bs  # LastBatchPolicy.DROP
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
