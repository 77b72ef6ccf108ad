"""Config surface: both schemas of the reference's ResNet-50 baseline compose to the same recipe; `_target_` aliases."""
import os

import pytest

from sota_imagenet_amd import config as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_defaults_match_reference_strict_config():
    cfg = C.compose(None, [])
    # sota_imagenet/arg_parser.py:13-156
    assert cfg.loader.batch_size == 256 and cfg.loader.image_size == 224 and cfg.val_loader.batch_size == 250
    assert cfg.bn_momentum == 0.1 and cfg.init_gamma == 1.72 and cfg.filter_from_wd is None
    assert cfg.optim._target_ == "torch.optim._multi_tensor.SGD" and cfg.optim.lr == 0 and cfg.optim.weight_decay == 1e-4
    assert cfg.run.accumulate_steps == 1 and cfg.run.fp16 is True and cfg.run.ema_decay == 0
    assert cfg.run.stages[0]["lr"] == [0.1, 0] and cfg.run.stages[0]["end"] == 90
    assert len(cfg.run.extra_callbacks) == 2


def test_hydra_schema_baseline():
    cfg = C.compose(os.path.join(ROOT, "configs", "resnet50_baseline.yaml"), [])
    assert cfg.model._target_ == "pytorch_tools.models.resnet50"
    assert cfg.optim.momentum == 0.9 and cfg.optim.weight_decay == 3e-5 and cfg.optim.lr == 0  # merged, not replaced
    assert cfg.criterion.smoothing == 0.1
    assert [(s["start"], s["end"], s["lr"], s["lr_mode"]) for s in cfg.run.stages] == [(0, 8, [0.001, 1.0], "linear"), (8, 90, [1.0, 0], "cos")]


def test_hydra_exp_cli_and_overrides():
    cfg = C.compose(None, ["+hydra_exp=1.r50_baseline", "loader.batch_size=128", "run.fp16=false", "debug=true"])
    assert cfg.log.exp_name == "r50_baseline" and cfg.loader.batch_size == 128 and cfg.run.fp16 is False and cfg.debug is True


def test_legacy_schema_maps_to_same_recipe():
    cfg = C.compose(os.path.join(ROOT, "configs", "resnet50_baseline_legacy.yaml"), [])
    assert cfg.model._target_ == "pytorch_tools.models.resnet50"
    assert cfg.optim.weight_decay == 3e-5 and cfg.optim.momentum == 0.9
    assert cfg.criterion.smoothing == 0.1
    assert cfg.loader.batch_size == 128 and cfg.loader.image_size == 224  # legacy recipe: bs 128
    assert [(s["start"], s["end"], s["lr"], s["lr_mode"]) for s in cfg.run.stages] == [(0, 8, [0, 1.0], "linear"), (8, 90, [1.0, 0], "cos")]


def test_legacy_progressive_resize_becomes_stage_extra_args():
    d = {"arch": "resnet50", "phases": [{"ep": 0, "sz": 128, "bs": 256}, {"ep": [0, 10], "lr": [0.0, 1.0]},
                                        {"ep": 6, "sz": 224, "bs": 128}, {"ep": [10, 20], "lr": [1.0, 0.0]}]}
    out = C.legacy_to_hydra(d)
    st = out["run"]["stages"]
    assert [(s["start"], s["end"]) for s in st] == [(0, 6), (6, 10), (10, 20)]
    assert st[1]["extra_args"] == {"image_size": 224, "batch_size": 128} and abs(st[0]["lr"][1] - 0.6) < 1e-12


def test_strictness_and_stage_validation():
    with pytest.raises(KeyError):
        C.compose(None, ["not_a_key=1"])
    with pytest.raises(AssertionError):
        C.compose(None, ["run.stages=[{start: 0, end: 5, lr: [0, 1]}, {start: 6, end: 9, lr: [1, 0]}]"])


def test_target_aliases_resolve_to_native_plugins():
    from sota_imagenet_amd import fit_wrapper, losses, models, optim

    assert C.resolve_target("pytorch_tools.models.resnet50") is models.resnet50
    assert C.resolve_target("pytorch_tools.losses.smooth.CrossEntropyLoss") is losses.CrossEntropyLoss
    assert C.resolve_target("torch.optim._multi_tensor.SGD") is optim.SGD
    assert isinstance(C.call({"_target_": "pytorch_tools.fit_wrapper.callbacks.Callback"}), fit_wrapper.Callback)
    crit = C.call({"_target_": "pytorch_tools.losses.smooth.CrossEntropyLoss", "smoothing": 0.1})
    assert crit.smoothing == 0.1


def test_bresnet50_encoder_legacy_recipe_maps_onto_the_plugin_surface():
    """BASELINE configs[3]: the legacy-schema recipe (reference configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-71)"""
    import os

    from sota_imagenet_amd import config as C

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = C.compose(os.path.join(root, "configs", "bresnet50_encoder_legacy.yaml"), [])
    m = C.to_plain(cfg.model)
    assert m["_target_"] == "pytorch_tools.models.resnet50" and m["stem_type"] == "deep" and m["antialias"] is True and m["attn_type"] == "eca"
    assert m["norm_act"] == "leaky_relu" and m["drop_rate"] == 0.2 and m["drop_connect_rate"] == 0.2
    assert cfg.weight_standardization is True and cfg.run.ema_decay == 0.9999 and cfg.criterion.smoothing == 0.1
    assert cfg.loader.image_size == 224 and cfg.loader.batch_size == 256 and cfg.val_loader.image_size == 288
    assert [(s["start"], s["end"]) for s in C.to_plain(cfg.run.stages)] == [(0, 8), (8, 100), (100, 150), (150, 200)]
    assert C.to_plain(cfg.run.extra_callbacks)[0]["_target_"].endswith("Cutmix")
    # the plugin builds the variant graph for these model_params (no GPU needed to construct it)
    model = C.call({k: v for k, v in m.items()}, weight_standardization=True, dtype="bf16")
    assert type(model).__name__ == "BResNet50" and abs(sum(p.numel() for p in model.parameters()) - 25.58e6) < 0.02e6
    assert all(c.standardize for c in model.modules() if type(c).__name__ == "_Conv") and model.drop_rate == 0.2
    # a variant recipe WITHOUT the flag / the drop rates gets what pytorch_tools gives: no standardisation, no dropout
    plain = C.call({k: v for k, v in m.items() if k not in ("drop_rate", "drop_connect_rate")}, dtype="bf16")
    assert not any(c.standardize for c in plain.modules() if type(c).__name__ == "_Conv")
    assert plain.drop_rate == 0.0 and plain.drop_connect_rate == 0.0
    off = C.call({k: v for k, v in m.items()}, weight_standardization=False, dtype="bf16")
    assert not any(c.standardize for c in off.modules() if type(c).__name__ == "_Conv")
