#!/usr/bin/env python3
"""train.py — same entry point and wiring as the reference's train.py, on the MI355X-native hot path.

    python train.py -c configs/resnet50_baseline.yaml [key.sub=value ...]
    python train.py +hydra_exp=1.r50_baseline loader.batch_size=128
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c configs/resnet50_baseline.yaml

Step for step it follows reference train.py:22-189: rank/world from the env (:27-29, arg_parser.py:151-152), run dir
`logs/<date>_<exp>/<HH-MM>` (configs/base.yaml:11-15), process group (:58-61), model via `_target_` (:64), BN momentum
patch (:76), criterion (:81), weight-decay filter + param groups (:83-89), optimizer with lr 0 (:92), resume (:98-109),
EMA before DDP (:111-114), stages -> PhasesScheduler (:117-131), the callback list in the reference's order (:129-143),
Runner (:145-152), data manager (:156), evaluate-only (:158-162), one `runner.fit` per stage (:164-173), final
`Acc@1 … Acc@5 …` line and `model_last.chpn` (:175-184).
What differs: DALI -> synthetic loader with the same contract; DDP -> flat-bucket RCCL all-reduce; `run.fp16` selects
bf16 activations (no loss scaler); hydra -> sota_imagenet_amd.config.
"""
import argparse
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sota_imagenet_amd import config as C  # noqa: E402
from sota_imagenet_amd import fit_wrapper as fw  # noqa: E402
from sota_imagenet_amd.data import SyntheticDataManager  # noqa: E402


def patch_bn_mom(model, momentum):
    """pt.utils.misc.patch_bn_mom (train.py:76)."""
    for m in model.modules():
        if hasattr(m, "momentum") and hasattr(m, "running_mean"):
            m.momentum = momentum


def filter_from_weight_decay(model, skip_list):
    """pt.utils.misc.filter_from_weight_decay (train.py:84): names containing a skip token (or 1-D tensors) get wd 0."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.dim() == 1 or any(tok in name for tok in skip_list):
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": decay}, {"params": no_decay, "weight_decay": 0.0}]


def count_parameters(model):
    return sum(p.numel() for p in model.parameters()), sum(p.numel() for p in model.parameters() if p.requires_grad)


class Logger:
    def __init__(self, master, path=None):
        self.master, self.path = master, path

    def info(self, msg):
        if not self.master:
            return
        line = time.strftime("[%m-%d %H:%M:%S] - ") + str(msg)
        print(line, flush=True)
        if self.path:
            with open(self.path, "a") as f:
                f.write(line + "\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-c", "--config", default=None, help="experiment YAML (Hydra or legacy schema)")
    ap.add_argument("overrides", nargs="*", help="+hydra_exp=<name> and dotted key=value overrides")
    args = ap.parse_args(argv)
    cfg = C.compose(args.config, args.overrides)

    start_time = time.time()
    cfg.distributed = cfg.world_size > 1
    cfg.is_master = cfg.local_rank == 0

    # run directory like hydra's: logs/<date>_<exp>/<HH-MM>  (configs/base.yaml:11-15)
    run_dir = os.path.join(ROOT, cfg.log.dir, time.strftime("%Y-%m-%d_") + cfg.log.exp_name, time.strftime("%H-%M"))
    if cfg.is_master:
        os.makedirs(run_dir, exist_ok=True)
        kwargs = {"universal_newlines": True, "stdout": subprocess.PIPE, "stderr": subprocess.DEVNULL, "cwd": ROOT}
        try:  # commit hash + diff for reproducibility (train.py:31-36)
            with open(os.path.join(run_dir, "commit_hash.txt"), "w") as f:
                f.write(subprocess.run(["git", "rev-parse", "--short", "HEAD"], **kwargs).stdout)
            with open(os.path.join(run_dir, "diff.txt"), "w") as f:
                f.write(subprocess.run(["git", "diff"], **kwargs).stdout)
        except OSError:
            pass
    logger = Logger(cfg.is_master, os.path.join(run_dir, "logs.txt"))
    import yaml

    logger.info("\n" + yaml.safe_dump(C.to_plain(cfg), sort_keys=False))
    logger.info(f"Torch version: {torch.__version__}")

    if not torch.cuda.is_available():
        raise SystemExit("train.py drives the MI355X hot path; no GPU is visible (there is no CPU fallback)")
    if cfg.random_seed is not None:
        torch.manual_seed(cfg.random_seed)
        import numpy as np

        np.random.seed(cfg.random_seed)
    torch.cuda.set_device(cfg.local_rank)
    if cfg.distributed:
        logger.info("Distributed initializing process group")
        torch.distributed.init_process_group(backend="nccl", init_method="env://", world_size=cfg.world_size)

    logger.info("Loading model")
    model_cfg = C.to_plain(cfg.model)
    if "dtype" not in model_cfg and C.resolve_target(model_cfg["_target_"]).__module__.startswith("sota_imagenet_amd"):
        model_cfg["dtype"] = "bf16" if cfg.run.fp16 else "fp32"  # the AMP flag of the reference (arg_parser.py:89-90)
    # the reference converts the built model only when the flag is set (conv_to_ws_conv, train.py:66-67); here standardisation is
    # a property of the variant graph's conv nodes, so the flag — on OR off — travels into the model factory
    own_model = C.resolve_target(model_cfg["_target_"]).__module__.startswith("sota_imagenet_amd")
    if cfg.weight_standardization and not own_model:
        raise NotImplementedError("weight standardisation needs one of this package's model plugins")
    if own_model:
        model_cfg["weight_standardization"] = bool(cfg.weight_standardization)
    model = C.call(model_cfg)
    if cfg.init_gamma is not None and hasattr(model, "reset_parameters"):
        model.reset_parameters(seed=cfg.random_seed or 0, gamma=cfg.init_gamma)
    model = model.cuda()
    patch_bn_mom(model, cfg.bn_momentum)
    if cfg.log.print_model:
        logger.info(model)

    criterion = C.call(cfg.criterion).cuda()
    if cfg.filter_from_wd is not None:
        opt_params = filter_from_weight_decay(model, skip_list=list(cfg.filter_from_wd))
    else:
        opt_params = [{"params": list(model.parameters())}]
    opt_params[0]["params"].extend(list(criterion.parameters()))
    optimizer = C.call(cfg.optim, opt_params)
    logger.info(f"Model params: {count_parameters(model)[0] / 1e6:.2f}M")

    if cfg.run.resume:
        resume_path = cfg.run.resume if os.path.isabs(cfg.run.resume) else os.path.join(ROOT, cfg.run.resume)
        checkpoint = torch.load(resume_path, map_location=f"cuda:{cfg.local_rank}")
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in checkpoint["state_dict"].items()}
        model.load_state_dict(sd, strict=False)
        logger.info(f"Loader model checkpoint from {resume_path}")
        if cfg.run.load_start_epoch:
            cfg.run.start_epoch = checkpoint["epoch"]
        try:
            optimizer.load_state_dict(checkpoint["optimizer"])
            logger.info("Loader optimizer state")
        except Exception:
            logger.info("Failed to load state dict into optimizer. It wasn't saved or optimizer has changed")

    ema_clb = fw.ModelEma(model, cfg.run.ema_decay) if cfg.run.ema_decay else fw.Callback()
    net = model
    if cfg.distributed:
        from sota_imagenet_amd.parallel import FlatBucketDDP

        net = FlatBucketDDP(model, device_ids=[cfg.local_rank])

    logger.info(C.to_plain(cfg.run.stages))
    lr_stages = []
    for stage in cfg.run.stages:
        if stage.get("lr") is None:
            continue
        lr_stages.append(dict(ep=(stage["start"], stage["end"]), lr=stage["lr"], mode=stage.get("lr_mode", "linear")))
    logger.info(f"Learning rate stages: {lr_stages}")

    callbacks = [
        fw.BatchMetrics([fw.Accuracy(), fw.Accuracy(5)]),
        fw.PhasesScheduler(lr_stages),
        fw.FileLogger(os.path.join(run_dir, "logs.txt")),
        fw.CheckpointSaver(run_dir, save_name="model.chpn", include_optimizer=cfg.log.save_optim),
        ema_clb,  # must come after the checkpoint saver (train.py:133-135)
        fw.Timer(),
        fw.ConsoleLogger(),
        fw.TensorBoard(run_dir, log_every=50),
        fw.Callback(),  # WeightDistributionTB is a research add-on outside the hot path
    ]
    callbacks += [C.call(clb_cfg) for clb_cfg in cfg.run.extra_callbacks]

    runner = fw.Runner(net, optimizer, criterion, callbacks=callbacks, use_fp16=cfg.run.fp16,
                       accumulate_steps=cfg.run.accumulate_steps)
    runner.state.random_seed = cfg.random_seed or 0

    data_manager = SyntheticDataManager(cfg)

    if cfg.run.evaluate:
        data_manager.set_stage(0)
        runner.callbacks.on_begin()
        runner.evaluate(data_manager.val_loader, steps=(None, 20)[cfg.debug])
        logger.info(fw._fmt_metrics("Val  ", runner.state.loss_meter.avg, runner.state.metric_meters))
        return runner.state.loss_meter.avg, runner.state.metric_meters

    for idx in range(len(data_manager)):
        data_manager.set_stage(idx)
        if data_manager.end_epoch <= cfg.run.start_epoch:
            continue
        runner.fit(
            data_manager.loader,
            steps_per_epoch=(None, 10)[cfg.debug],
            val_loader=data_manager.val_loader,
            val_steps=(None, 20)[cfg.debug],
            epochs=data_manager.end_epoch,
            start_epoch=max(data_manager.start_epoch, cfg.run.start_epoch),
        )
    logger.info(f"Model params: {count_parameters(model)[0] / 1e6:.2f}M")
    metrics = runner.state.val_metrics
    logger.info(f"Acc@1 {metrics['Acc@1'].avg:.3f} Acc@5 {metrics['Acc@5'].avg:.3f}")
    m = (time.time() - start_time) / 60
    logger.info(f"Total time: {int(m / 60)}h {m % 60:.1f}m")
    if cfg.is_master:
        torch.save(model.state_dict(), os.path.join(run_dir, "model_last.chpn"))
    if cfg.distributed:
        torch.distributed.destroy_process_group()
    return runner.state.val_loss.avg, metrics


if __name__ == "__main__":
    main()
