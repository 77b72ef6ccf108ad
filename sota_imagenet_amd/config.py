"""Config surface: the reference's Hydra `StrictConfig` re-stated with PyYAML (hydra / omegaconf are not installed).

Accepts both schemas the reference has carried for the ResNet-50 baseline, so `configs/resnet50_baseline.yaml` is a
drop-in in either form:
  * current Hydra schema — sota_imagenet/arg_parser.py:13-156 (every field + default below), experiment files such as
    configs/hydra_exp/1.r50_baseline.yaml:19-44 (`# @package _global_`, `defaults: [/base@_here_]` are tolerated),
    CLI `+hydra_exp=<name>` and dotted overrides `key.sub=value` (docker/slurm_train_1gpu.slurm:10);
  * legacy flat schema — configs/_old_configs/_first_attempts/resnet50_baseline.yaml:5-13
    (`arch`, `weight_decay`, `smooth`, `phases: [...]`), mapped per SURVEY.md §A.2.
`_target_` dicts are instantiated by `call()` like hydra.utils.call (train.py:64,81,92,143); the three hot-path
targets of the reference resolve to this package's native plugins through TARGET_ALIASES.
"""
import copy
import importlib
import os
import re

import yaml

TARGET_ALIASES = {
    # reference plugin target                          -> MI355X-native implementation
    "pytorch_tools.models.resnet50": "sota_imagenet_amd.models.resnet50",
    "pytorch_tools.losses.smooth.CrossEntropyLoss": "sota_imagenet_amd.losses.CrossEntropyLoss",
    "pytorch_tools.losses.CrossEntropyLoss": "sota_imagenet_amd.losses.CrossEntropyLoss",
    "torch.optim._multi_tensor.SGD": "sota_imagenet_amd.optim.SGD",
    "torch.optim.SGD": "sota_imagenet_amd.optim.SGD",
    "pytorch_tools.fit_wrapper.callbacks.Callback": "sota_imagenet_amd.fit_wrapper.Callback",
    "pytorch_tools.fit_wrapper.callbacks.Cutmix": "sota_imagenet_amd.callbacks.Cutmix",
    "pytorch_tools.fit_wrapper.callbacks.Mixup": "sota_imagenet_amd.callbacks.Mixup",
    "sota_imagenet.callbacks.CutmixMixup": "sota_imagenet_amd.callbacks.CutmixMixup",
}


def default_config():
    """StrictConfig defaults — sota_imagenet/arg_parser.py:13-156."""
    loader = dict(image_size=224, batch_size=256, workers=6, num_classes=1000, _is_train=True,
                  root_data_dir=os.environ.get("IMAGENET_DIR", ""), use_tfrecords=False, min_area=0.08, blur_prob=0,
                  gray_prob=0, color_twist_prob=0, contrast_range=[0.7, 1.3], brightness_range=[0.7, 1.3],
                  random_interpolation=False, re_prob=0, re_count=3)
    val_loader = dict(image_size=224, batch_size=250, workers=6, num_classes=1000, _is_train=False,
                      root_data_dir=os.environ.get("IMAGENET_DIR", ""), use_tfrecords=False, full_crop=False)
    return dict(
        loader=loader,
        val_loader=val_loader,
        model=dict(_target_="pytorch_tools.models.resnet18"),
        weight_standardization=False,
        filter_from_wd=None,
        bn_momentum=0.1,
        init_gamma=1.72,
        optim=dict(_target_="torch.optim._multi_tensor.SGD", lr=0, weight_decay=1e-4),
        criterion=dict(_target_="pytorch_tools.losses.smooth.CrossEntropyLoss"),
        run=dict(stages=[dict(start=0, end=90, lr=[0.1, 0], lr_mode="linear", extra_args=None)], resume=None,
                 load_start_epoch=True, start_epoch=0, accumulate_steps=1, ema_decay=0, fp16=True,
                 extra_callbacks=[dict(_target_="pytorch_tools.fit_wrapper.callbacks.Callback"),
                                  dict(_target_="pytorch_tools.fit_wrapper.callbacks.Callback")],
                 evaluate=False),
        log=dict(exp_name="test_run", dir="logs", print_model=False, histogram=False, save_optim=False),
        debug=False,
        random_seed=None,
        world_size=int(os.environ.get("WORLD_SIZE", 1)),
        local_rank=int(os.environ.get("LOCAL_RANK", 0)),
        distributed=False,
        is_master=True,
        # additions of this framework (not in the reference schema)
        data=dict(synthetic=True, train_size=1281167, val_size=50000, pool=8),
    )


class Cfg(dict):
    """dict with attribute access (cfg.run.stages) like an OmegaConf node."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(x):
    if isinstance(x, dict):
        return Cfg({k: to_cfg(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [to_cfg(v) for v in x]
    return x


def to_plain(x):
    if isinstance(x, dict):
        return {k: to_plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_plain(v) for v in x]
    return x


def deep_merge(base, over):
    """Hydra-style merge: dicts merge recursively (experiment files ADD keys to optim/model/criterion), the rest replaces."""
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            deep_merge(base[k], v)
        else:
            base[k] = copy.deepcopy(v)
    return base


_ENV = re.compile(r"\$\{(?:oc\.)?env:([A-Za-z_][A-Za-z0-9_]*)(?:,([^}]*))?\}")


def _interp(x):
    if isinstance(x, str):
        m = _ENV.fullmatch(x.strip())
        if m:
            return _yaml(os.environ.get(m.group(1), m.group(2) or "")) if (m.group(1) in os.environ or m.group(2)) else None
        return _ENV.sub(lambda m: os.environ.get(m.group(1), m.group(2) or ""), x)
    if isinstance(x, dict):
        return {k: _interp(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_interp(v) for v in x]
    return x


def is_legacy(d):
    return "arch" in d or "phases" in d


def legacy_to_hydra(d):
    """legacy flat keys -> Hydra-schema overrides (data phases have an int `ep`, schedule phases a pair)."""
    out = {"log": {}, "optim": {}, "criterion": {}, "loader": {}, "run": {}}
    if "name" in d:
        out["log"]["exp_name"] = d["name"]
    if "arch" in d:
        out["model"] = dict(_target_=f"pytorch_tools.models.{d['arch']}", **(d.get("model_params") or {}))
    if "weight_decay" in d:
        out["optim"]["weight_decay"] = float(d["weight_decay"])
    if d.get("smooth"):
        out["criterion"]["smoothing"] = 0.1
    if "ema_decay" in d:
        out["run"]["ema_decay"] = d["ema_decay"]
    for k in ("weight_standardization", "resume", "accumulate_steps"):
        if k in d:
            (out if k == "weight_standardization" else out["run"])[k] = d[k]
    if d.get("no_bn_wd"):
        out["filter_from_wd"] = ["bn", "bias"]
    if d.get("ctwist"):
        out["loader"]["color_twist_prob"] = 0.5
    stages, data_phases = [], []
    for ph in d.get("phases", []):
        ep = ph["ep"]
        if isinstance(ep, (list, tuple)):
            stages.append(dict(start=ep[0], end=ep[1], lr=list(ph["lr"]) if isinstance(ph["lr"], (list, tuple)) else [ph["lr"], ph["lr"]],
                               lr_mode=ph.get("mode", "linear"), extra_args=None))
            if "mom" in ph:
                out["optim"]["momentum"] = ph["mom"] if not isinstance(ph["mom"], (list, tuple)) else ph["mom"][0]
        else:
            data_phases.append(ph)
    for ph in data_phases:
        extra = {}
        if "sz" in ph:
            extra["image_size"] = ph["sz"]
        if "bs" in ph:
            extra["batch_size"] = ph["bs"]
        if "val_sz" in ph:  # e.g. BResNet50_encoder.yaml:63: validate on larger images than trained on
            out.setdefault("val_loader", {})["image_size"] = ph["val_sz"]
        if ph["ep"] == 0:
            out["loader"].update(extra)
        else:  # a later data phase becomes extra_args of the stage that starts there (split the stage if needed)
            new = []
            for st in stages:
                if st["start"] < ph["ep"] < st["end"]:
                    span = st["end"] - st["start"]
                    mid_lr = st["lr"][0] + (st["lr"][1] - st["lr"][0]) * (ph["ep"] - st["start"]) / span
                    new.append(dict(st, end=ph["ep"], lr=[st["lr"][0], mid_lr]))
                    new.append(dict(st, start=ph["ep"], lr=[mid_lr, st["lr"][1]], extra_args=dict(extra)))
                elif st["start"] == ph["ep"]:
                    new.append(dict(st, extra_args=dict(extra)))
                else:
                    new.append(st)
            stages = new
    if stages:
        out["run"]["stages"] = stages
    if d.get("cutmix"):
        out["run"]["extra_callbacks"] = [dict(_target_="pytorch_tools.fit_wrapper.callbacks.Cutmix", alpha=float(d["cutmix"]),
                                              num_classes=1000, prob=0.5)]
    return {k: v for k, v in out.items() if v not in ({}, None)}


class _Loader(yaml.SafeLoader):
    """SafeLoader + YAML-1.2 style floats (`3e-5`, `1e-4`), which OmegaConf accepts and the reference configs use."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*)(?:\.[0-9_]*)?[eE][-+]?[0-9]+$"),
    list("-+0123456789"),
)


def _yaml(text):
    return yaml.load(text, Loader=_Loader)


def load_yaml(path):
    with open(path) as f:
        d = _yaml(f.read()) or {}
    d.pop("defaults", None)  # hydra composition header (`- /base@_here_`)
    d.pop("hydra", None)
    return d


def _set_dotted(cfg, dotted, value):
    node = cfg
    keys = dotted.split(".")
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = value


def compose(config_path=None, overrides=(), config_dir=None):
    """defaults <- base.yaml (if present) <- experiment YAML (either schema) <- CLI overrides.  Returns a Cfg."""
    cfg = default_config()
    config_dir = config_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
    base = os.path.join(config_dir, "base.yaml")
    if os.path.exists(base):
        deep_merge(cfg, load_yaml(base))
    files = [config_path] if config_path else []
    rest = []
    for ov in overrides:
        m = re.fullmatch(r"\+?hydra_exp=(.+)", ov)
        if m:
            name = m.group(1)
            cand = [os.path.join(config_dir, "hydra_exp", name + ".yaml"), os.path.join(config_dir, name + ".yaml"), name]
            files.append(next((c for c in cand if os.path.exists(c)), cand[0]))
        else:
            rest.append(ov)
    for f in files:
        d = load_yaml(f)
        deep_merge(cfg, legacy_to_hydra(d) if is_legacy(d) else d)
    for ov in rest:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not key=value")
        k, v = ov.lstrip("+").split("=", 1)
        _set_dotted(cfg, k, _yaml(v))
    cfg = _interp(cfg)
    cfg["world_size"] = int(cfg.get("world_size") or os.environ.get("WORLD_SIZE", 1))
    cfg["local_rank"] = int(cfg.get("local_rank") or os.environ.get("LOCAL_RANK", 0))
    for st in cfg["run"]["stages"]:
        st.setdefault("lr_mode", "linear")
        st.setdefault("extra_args", None)
        st.setdefault("lr", None)
    validate(cfg)
    return to_cfg(cfg)


def validate(cfg):
    """stage list must be contiguous — sota_imagenet/dali_dataloader.py:206-211."""
    end = cfg["run"]["stages"][0]["start"] if cfg["run"]["stages"] else 0
    for st in cfg["run"]["stages"]:
        assert st["start"] == end, "error in data stages. start != end"
        assert st["end"] > st["start"], "error in data stages, end <= start"
        end = st["end"]
    unknown = set(cfg.keys()) - set(default_config().keys())
    if unknown:
        raise KeyError(f"unknown top-level config keys: {sorted(unknown)} (strict config, arg_parser.py:121-156)")


def resolve_target(path):
    path = TARGET_ALIASES.get(path, path)
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def call(node, *args, **extra):
    """hydra.utils.call: instantiate `_target_` with the node's remaining keys as kwargs (positional args first)."""
    node = to_plain(node)
    kwargs = {k: v for k, v in node.items() if k != "_target_"}
    kwargs.update(extra)
    return resolve_target(node["_target_"])(*args, **kwargs)
