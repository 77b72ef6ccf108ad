"""Synthetic data feed with the reference loader's protocol and tensor contract.

Replaces sota_imagenet/dali_dataloader.py for the hot path (north_star: "the DALI pipeline is swapped for a synthetic
NCHW->NHWC generator"): `SyntheticLoader` mirrors `DaliLoader` (:163-186 — `batch_size` property, `__len__ =
ceil(size / batch)`, `__iter__` yielding `(data, label)`; last partial batch dropped :175; shard = rank :47) and emits
what the DALI pipeline emits: data float32 NCHW `[N,3,S,S]` normalised with mean 127.5 / std 51 (:27-29, :113-122)
and one-hot float labels `[N,num_classes]` (:123), both already on the GPU.  `SyntheticDataManager` mirrors
`DaliDataManager` (:189-239): `set_stage(i)`, `.loader`, `.val_loader`, `.start_epoch/.end_epoch`, `len()`; loaders are
rebuilt only when a stage carries `extra_args` (progressive resize).  The NCHW->NHWC conversion happens inside the
native ingest kernel, so the CPU oracle and the GPU path consume identical tensors.
"""
import math
from copy import deepcopy

import torch

from .fit_wrapper import env_rank, env_world_size
from .synth import synthetic_batch


class SyntheticLoader:
    def __init__(self, cfg, size, seed=0, device=None, pool=8, is_val=False):
        self.cfg = cfg
        self._bs = int(cfg["batch_size"])
        self.image_size = int(cfg["image_size"])
        self.num_classes = int(cfg.get("num_classes", 1000))
        self.rank, self.world = env_rank(), env_world_size()
        self._size = int(math.ceil(size / self.world))  # this shard's share of the data set
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.seed = int(seed or 0) + (500 if is_val else 0)
        self._pool_n = max(1, int(pool))
        self._pool = None

    @property
    def batch_size(self):
        return self._bs

    def __len__(self):
        return math.ceil(self._size / self._bs)

    def _build_pool(self):
        self._pool = [synthetic_batch(self._bs, self.image_size, self.num_classes, seed=self.seed, stream=self.rank,
                                      index=i, device=self.device) for i in range(self._pool_n)]

    def __iter__(self):
        if self._pool is None:
            self._build_pool()  # generation cost stays off the step clock; the pool is cycled
        n_full = self._size // self._bs  # LastBatchPolicy.DROP
        for i in range(max(n_full, 1)):
            yield self._pool[i % self._pool_n]


def make_loader(cfg, size, seed, device, pool, is_val, source="auto"):
    """`data.source`: "synthetic" (the benchmark feed), "folder" (JPEG folders under loader.root_data_dir/{train,val}, resized
    on the GPU — image_loader.py), or "auto": folders when that directory exists, synthetic otherwise (no IMAGENET_DIR in
    the benchmark environment)."""
    import os

    sub = os.path.join(str(cfg.get("root_data_dir") or ""), "val" if is_val else "train")
    if source == "folder" or (source == "auto" and cfg.get("root_data_dir") and os.path.isdir(sub)):
        from .image_loader import ImageFolderLoader

        return ImageFolderLoader(cfg, is_val=is_val, seed=seed, device=device)
    if source not in ("auto", "synthetic"):
        raise ValueError(f"data.source = {source!r}: expected synthetic | folder | auto")
    return SyntheticLoader(cfg, size, seed, device, pool, is_val=is_val)


class SyntheticDataManager:
    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.stages = cfg.run.stages
        self.tot_epochs = max(st["end"] for st in self.stages)
        self._validate_stages()
        self.device = device
        self.loader = None
        self.val_loader = None
        self.start_epoch = None
        self.end_epoch = None

    def __len__(self):
        return len(self.stages)

    def _validate_stages(self):
        end = self.stages[0]["start"] if self.stages else 0
        for st in self.stages:
            assert st["start"] == end, "error in data stages. start != end"
            assert st["end"] > st["start"], "error in data stages, end <= start"
            end = st["end"]

    def set_stage(self, idx):
        st = self.stages[idx]
        self.start_epoch, self.end_epoch = st["start"], st["end"]
        if st.get("extra_args") is None and self.loader is not None:
            return  # only the learning rate changed
        train_cfg = deepcopy(dict(self.cfg.loader))
        val_cfg = deepcopy(dict(self.cfg.val_loader))
        for k, v in (st.get("extra_args") or {}).items():
            train_cfg[k] = v
        val_cfg["image_size"] = train_cfg["image_size"]  # "for now only image size changes in val loader"
        if self.loader is not None:
            del self.loader, self.val_loader
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        d = self.cfg.get("data", {})
        seed = self.cfg.get("random_seed") or 0
        self.loader = make_loader(train_cfg, d.get("train_size", 1281167), seed, self.device, d.get("pool", 8), False, d.get("source", "auto"))
        self.val_loader = make_loader(val_cfg, d.get("val_size", 50000), seed, self.device, min(d.get("pool", 8), 4), True, d.get("source", "auto"))


# the reference's name for the same object (sota_imagenet/dali_dataloader.py:189): configs / scripts that import it keep working
DaliDataManager = DataManager = SyntheticDataManager
