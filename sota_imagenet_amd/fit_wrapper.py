"""Runner + callbacks: the loop surface of `pytorch_tools.fit_wrapper` that the reference's train.py drives.

Same constructor / `fit` / `evaluate` / `state` / callback-hook surface as the reference uses:
  Runner(model, optimizer, criterion, callbacks=[...], use_fp16=..., accumulate_steps=...)   train.py:145-152
  runner.fit(loader, steps_per_epoch=, val_loader=, val_steps=, epochs=, start_epoch=)       train.py:166-173
  runner.evaluate(loader); runner.state.loss_meter.avg / .metric_meters / .val_metrics       train.py:158-162,178
  Callback hooks on_begin/on_end/on_epoch_begin/on_epoch_end/on_loader_begin/on_loader_end/
  on_batch_begin/on_batch_end/on_after_backward           sota_imagenet/callbacks.py:15,38,74,80,241,288
  inner step = criterion(model(data), target) -> backward -> (every accumulate_steps) optimizer.step, zero_grad
                                                            re-enacted by the reference in callbacks.py:314-317
The package itself is not vendored in the reference tree; behaviour follows SURVEY.md Appendix C and the call sites.

MI355X-specific choices: the loop never reads a device scalar per step (the reference syncs for its meters every
batch) — loss / metric sums accumulate on the device and are read when `.avg` is asked for; `use_fp16` has no loss
scaler because the low-precision mode of this path is bf16 (selected on the model: resnet50(dtype="bf16")).
"""
import math
import os
import time
from collections import OrderedDict
from copy import deepcopy

import torch


# ---------------------------------------------------------------------------------------------------------------
def env_rank():
    return int(os.environ.get("RANK", 0))


def env_world_size():
    return int(os.environ.get("WORLD_SIZE", 1))


def rank_zero_only(fn):
    """sota_imagenet/callbacks.py:11 uses pt_clb.rank_zero_only on side-effect hooks."""

    def wrapped(*a, **k):
        if env_rank() == 0:
            return fn(*a, **k)
        return None

    return wrapped


class AverageMeter:
    """running average that accepts device scalars without synchronising until `.avg` is read."""

    def __init__(self, name="meter"):
        self.name = name
        self.reset()

    def reset(self):
        self._sum = 0.0
        self._tsum = None
        self.count = 0
        self.val = 0.0

    def update(self, val, n=1):
        if torch.is_tensor(val):
            v = val.detach().float()
            self._tsum = v * n if self._tsum is None else self._tsum + v * n
            self.val = v
        else:
            self._sum += float(val) * n
            self.val = float(val)
        self.count += n

    @property
    def sum(self):
        s = self._sum
        if self._tsum is not None:
            s += float(self._tsum.item())
        return s

    @property
    def avg(self):
        return self.sum / max(self.count, 1)

    def __call__(self):
        return self.avg


class Accuracy:
    """pt.metrics.Accuracy(topk): percent of rows whose target class is among the top-k logits; 2-D (one-hot / soft)
    targets are reduced with argmax (train.py:130; loader emits one-hot rows, dali_dataloader.py:123)."""

    def __init__(self, topk=1):
        self.topk = topk
        self.name = f"Acc@{topk}"

    @torch.no_grad()
    def __call__(self, output, target):
        if target.dim() == 2:
            target = target.argmax(1)
        _, pred = output.topk(self.topk, dim=1)
        return (pred == target.view(-1, 1)).any(1).float().mean() * 100.0


class RunnerState:
    def __init__(self, model=None, optimizer=None, criterion=None, use_fp16=False, accumulate_steps=1):
        self.model = model
        self.optimizer = optimizer
        self.criterion = criterion
        self.use_fp16 = use_fp16
        self.accumulate_steps = accumulate_steps
        self.grad_scaler = None  # bf16 path: no loss scaling (reference: GradScaler, callbacks.py:295,309)
        self.is_train = True
        self.epoch = 0
        self.num_epochs = 1
        self.epoch_size = None
        self.step = None
        self.batch_size = 0
        self.global_sample_step = 0
        self.input = None
        self.output = None
        self.tb_logger = None
        self.loss_meter = AverageMeter("loss")
        self.metric_meters = OrderedDict()
        self.train_loss = None
        self.train_metrics = None
        self.val_loss = None
        self.val_metrics = None
        self.epoch_log = 0
        self.world_size = env_world_size()
        self.rank = env_rank()
        self.random_seed = 0  # train.py sets cfg.random_seed: device-side samplers (CutmixMixup) derive their stream from (seed, rank)


class Callback:
    """no-op base; also the placeholder the reference instantiates for disabled features (train.py:12,112,140)."""

    def __init__(self):
        self.state = RunnerState()

    def set_state(self, state):
        self.state = state

    def on_begin(self): pass
    def on_end(self): pass
    def on_epoch_begin(self): pass
    def on_epoch_end(self): pass
    def on_loader_begin(self): pass
    def on_loader_end(self): pass
    def on_batch_begin(self): pass
    def on_batch_end(self): pass
    def on_after_backward(self): pass


class Callbacks(Callback):
    def __init__(self, callbacks):
        super().__init__()
        if callbacks is None:
            callbacks = []
        self.callbacks = list(callbacks) if isinstance(callbacks, (list, tuple)) else [callbacks]

    def set_state(self, state):
        self.state = state
        for c in self.callbacks:
            c.set_state(state)

    def _all(self, name):
        for c in self.callbacks:
            getattr(c, name)()

    def on_begin(self): self._all("on_begin")
    def on_end(self): self._all("on_end")
    def on_epoch_begin(self): self._all("on_epoch_begin")
    def on_epoch_end(self): self._all("on_epoch_end")
    def on_loader_begin(self): self._all("on_loader_begin")
    def on_loader_end(self): self._all("on_loader_end")
    def on_batch_begin(self): self._all("on_batch_begin")
    def on_batch_end(self): self._all("on_batch_end")
    def on_after_backward(self): self._all("on_after_backward")


# ---------------------------------------------------------------------------------------------------------------
class BatchMetrics(Callback):
    """train.py:130 — pt_clb.BatchMetrics([Accuracy(), Accuracy(5)])."""

    def __init__(self, metrics):
        super().__init__()
        self.metrics = list(metrics) if isinstance(metrics, (list, tuple)) else [metrics]
        self.metric_names = [m.name for m in self.metrics]

    def on_begin(self):
        for name in self.metric_names:
            self.state.metric_meters[name] = AverageMeter(name)

    @torch.no_grad()
    def on_batch_end(self):
        _, target = self.state.input
        out = self.state.output
        for metric, name in zip(self.metrics, self.metric_names):
            self.state.metric_meters[name].update(metric(out, target))


def phase_lr(phase, epoch, step, epoch_size):
    """learning rate of one scheduler phase {ep:(start,end), lr:(a,b)|scalar, mode}; SURVEY.md Appendix C formulae."""
    start, end = phase["ep"]
    lr = phase["lr"]
    a, b = (lr[0], lr[-1]) if isinstance(lr, (list, tuple)) else (lr, lr)
    pct = ((epoch - start) * epoch_size + step) / float(max(end - start, 1e-12) * epoch_size)
    mode = phase.get("mode", "linear")
    if mode == "cos":
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    if mode == "poly":
        return b + (a - b) * (1.0 - pct) ** 2
    return a + (b - a) * pct


class PhasesScheduler(Callback):
    """train.py:117-131 — per-batch LR (and optional momentum) from `lr_stages` dicts; writes every param_group."""

    def __init__(self, phases):
        super().__init__()
        self.phases = [dict(p) for p in phases]
        for p in self.phases:
            ep = p["ep"]
            p["ep"] = (ep, ep) if not isinstance(ep, (list, tuple)) else (ep[0], ep[-1])
        self.current = None

    def _pick(self, epoch):
        cur = None
        for p in self.phases:
            if p["ep"][0] <= epoch:
                cur = p
        return cur

    def on_epoch_begin(self):
        self.current = self._pick(self.state.epoch)

    def on_batch_begin(self):
        if not self.state.is_train or self.current is None:
            return
        lr = phase_lr(self.current, self.state.epoch, self.state.step, self.state.epoch_size)
        for g in self.state.optimizer.param_groups:
            g["lr"] = lr
            if "mom" in self.current:
                mom = self.current["mom"]
                g["momentum"] = mom if not isinstance(mom, (list, tuple)) else mom[0]


class Timer(Callback):
    """train.py:137 — wall-clock bookkeeping; here it also yields images/sec (the BASELINE metric)."""

    def __init__(self):
        super().__init__()
        self.images_per_sec = 0.0

    def on_loader_begin(self):
        self._t0 = time.time()
        self._n0 = self.state.global_sample_step

    def on_loader_end(self):
        if self.state.is_train:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = max(time.time() - self._t0, 1e-9)
            self.images_per_sec = (self.state.global_sample_step - self._n0) / dt
            self.state.images_per_sec = self.images_per_sec


def _fmt_metrics(prefix, loss, metrics):
    parts = [f"{prefix} loss: {loss:.4f}"]
    for k, m in (metrics or {}).items():
        parts.append(f"{k}: {m.avg:.4f}")
    return " | ".join(parts)


class ConsoleLogger(Callback):
    """train.py:138 — `Train loss: … | Acc@1: … | Acc@5: …` lines (format of configs/hydra_exp/1.r50_baseline.yaml:9-10)."""

    def __init__(self, sink=print):
        super().__init__()
        self.sink = sink

    @rank_zero_only
    def on_epoch_end(self):
        ips = getattr(self.state, "images_per_sec", None)
        head = f"Epoch {self.state.epoch + 1:3d}/{self.state.num_epochs}" + (f" | {ips:.1f} img/s" if ips else "")
        self.sink(head)
        self.sink(_fmt_metrics("Train", self.state.train_loss.avg, self.state.train_metrics))
        if self.state.val_loss is not None:
            self.sink(_fmt_metrics("Val  ", self.state.val_loss.avg, self.state.val_metrics))


class FileLogger(ConsoleLogger):
    """train.py:132 — same lines appended to a log file (the reference routes them through loguru into logs.txt)."""

    def __init__(self, path="logs.txt"):
        super().__init__(sink=self._write)
        self.path = path

    def _write(self, line):
        with open(self.path, "a") as f:
            f.write(time.strftime("[%m-%d %H:%M:%S] - ") + line + "\n")


class TensorBoard(Callback):
    """train.py:139 — scalar logging every `log_every` steps when a SummaryWriter is importable; otherwise inert."""

    def __init__(self, log_dir, log_every=50):
        super().__init__()
        self.log_dir = log_dir
        self.log_every = log_every
        self.writer = None

    @rank_zero_only
    def on_begin(self):
        try:
            from torch.utils.tensorboard import SummaryWriter

            self.writer = SummaryWriter(self.log_dir)
            self.state.tb_logger = self.writer
        except Exception:
            self.writer = None

    @rank_zero_only
    def on_batch_end(self):
        if self.writer is None or not self.state.is_train or self.state.step % self.log_every:
            return
        self.writer.add_scalar("train_/loss", float(self.state.loss_meter.val), self.state.global_sample_step)
        self.writer.add_scalar("train_/lr", self.state.optimizer.param_groups[0]["lr"], self.state.global_sample_step)

    @rank_zero_only
    def on_end(self):
        if self.writer is not None:
            self.writer.close()


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


class CheckpointSaver(Callback):
    """train.py:133-135 — saves {"epoch", "state_dict"[, "optimizer"]} when the monitored value improves
    (keys read back at train.py:98-109)."""

    def __init__(self, save_dir, save_name="model_{ep}_{metric:.2f}.chpn", monitor="loss", mode="min", include_optimizer=False):
        super().__init__()
        self.save_dir = save_dir
        self.save_name = save_name
        self.monitor = monitor
        self.include_optimizer = include_optimizer
        self.best = math.inf if mode == "min" else -math.inf
        self.better = (lambda a, b: a < b) if mode == "min" else (lambda a, b: a > b)

    def _current(self):
        if self.monitor == "loss":
            src = self.state.val_loss if self.state.val_loss is not None else self.state.train_loss
            return src.avg
        src = self.state.val_metrics if self.state.val_metrics is not None else self.state.train_metrics
        return src[self.monitor].avg

    @rank_zero_only
    def on_epoch_end(self):
        cur = self._current()
        if not self.better(cur, self.best):
            return
        self.best = cur
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, self.save_name.format(ep=self.state.epoch + 1, metric=cur))
        ckpt = {"epoch": self.state.epoch + 1, "state_dict": _unwrap(self.state.model).state_dict()}
        if self.include_optimizer:
            ckpt["optimizer"] = self.state.optimizer.state_dict()
        torch.save(ckpt, path)


class ModelEma(Callback):
    """train.py:111-112 — exponential moving average of the flat parameter / buffer arrays (one fused lerp each);
    EMA weights are swapped in for validation and checkpointing, hence the ordering constraint at train.py:133-135."""

    def __init__(self, model, decay=0.9999):
        super().__init__()
        self.decay = decay
        self.model = _unwrap(model)
        self._flat = hasattr(self.model, "flat_params")
        if self._flat:
            self.ema = [self.model.flat_params.detach().clone(), self.model._flat_buffers.detach().clone()]
        else:
            self.ema = [t.detach().clone() for t in self.model.state_dict().values()]
        self._swapped = False
        self._fused = False  # the optimizer's step kernel advances ema[0] (optim.SGD.attach_ema)

    def on_begin(self):
        # one optimizer step per batch and a flat-array model under the native SGD: the parameter average moves inside the step kernel
        # (mi355_sgd_step_ema), this callback keeps the buffers (BN running statistics) only
        opt = getattr(self.state, "optimizer", None)
        if self._flat and hasattr(opt, "attach_ema") and getattr(self.state, "accumulate_steps", 1) == 1 and self.model.flat_params.is_cuda:
            opt.attach_ema(self.model.flat_params, self.ema[0], self.decay)
            self._fused = True

    def _live(self):
        if self._flat:
            return [self.model.flat_params, self.model._flat_buffers]
        return list(self.model.state_dict().values())

    @torch.no_grad()
    def on_batch_end(self):
        if not self.state.is_train:
            return
        for k, (e, p) in enumerate(zip(self.ema, self._live())):
            if k == 0 and self._fused:
                continue
            if e.dtype.is_floating_point:
                e.lerp_(p.detach(), 1.0 - self.decay)
            else:
                e.copy_(p)

    @torch.no_grad()
    def _swap(self):
        for e, p in zip(self.ema, self._live()):
            tmp = p.detach().clone()
            p.copy_(e)
            e.copy_(tmp)
        self._swapped = not self._swapped

    def on_loader_begin(self):
        if not self.state.is_train and not self._swapped:
            self._swap()  # validate with EMA weights

    def on_epoch_end(self):
        pass  # CheckpointSaver (placed before this callback) has already saved the EMA weights

    def on_epoch_begin(self):
        if self._swapped:
            self._swap()  # back to the live weights for training

    def on_end(self):
        if self._swapped:
            self._swap()


# ---------------------------------------------------------------------------------------------------------------
class Runner:
    def __init__(self, model, optimizer, criterion, callbacks=None, gradient_clip_val=None, use_fp16=False, accumulate_steps=1):
        self.state = RunnerState(model=model, optimizer=optimizer, criterion=criterion, use_fp16=use_fp16,
                                 accumulate_steps=accumulate_steps)
        self.callbacks = Callbacks(callbacks)
        self.callbacks.set_state(self.state)
        self.gradient_clip_val = gradient_clip_val
        if hasattr(optimizer, "attach_model") and hasattr(_unwrap(model), "mark_grads_clean"):
            optimizer.attach_model(_unwrap(model))

    def fit(self, train_loader, steps_per_epoch=None, val_loader=None, val_steps=None, epochs=1, start_epoch=0):
        self.state.num_epochs = epochs
        self.state.batch_size = getattr(train_loader, "batch_size", 1)
        m = _unwrap(self.state.model)
        if hasattr(m, "reseed"):  # device-side drop-connect / dropout: one stream per (run seed, rank), like CutmixMixup
            m.reseed(self.state.random_seed, self.state.rank)
        self.callbacks.on_begin()
        for epoch in range(start_epoch, epochs):
            self.state.is_train = True
            self.state.epoch = epoch
            self.state.epoch_log = epoch + 1
            self.callbacks.on_epoch_begin()
            self.state.model.train()
            self._run_loader(train_loader, steps=steps_per_epoch)
            self.state.train_loss = deepcopy(self.state.loss_meter)
            self.state.train_metrics = deepcopy(self.state.metric_meters)
            if val_loader is not None:
                self.evaluate(val_loader, steps=val_steps)
                self.state.val_loss = deepcopy(self.state.loss_meter)
                self.state.val_metrics = deepcopy(self.state.metric_meters)
            self.callbacks.on_epoch_end()
        self.callbacks.on_end()

    def evaluate(self, loader, steps=None):
        self.state.is_train = False
        self.state.model.eval()
        self._run_loader(loader, steps=steps)
        return self.state.loss_meter.avg, [m.avg for m in self.state.metric_meters.values()]

    def _make_step(self):
        data, target = self.state.input
        output = self.state.model(data)
        loss = self.state.criterion(output, target)
        self.state.output = output
        if self.state.is_train:
            last_micro = (self.state.step + 1) % self.state.accumulate_steps == 0
            no_sync = getattr(self.state.model, "no_sync", None)
            if no_sync is not None and not last_micro:  # data parallel: reduce the accumulated gradients once, on the last micro-step
                with no_sync():
                    (loss / self.state.accumulate_steps).backward()
            else:
                (loss / self.state.accumulate_steps).backward()
            self.callbacks.on_after_backward()
            if (self.state.step + 1) % self.state.accumulate_steps == 0:
                if self.gradient_clip_val is not None:
                    torch.nn.utils.clip_grad_norm_(self.state.model.parameters(), self.gradient_clip_val)
                self.state.optimizer.step()
                self.state.optimizer.zero_grad()
            self.state.global_sample_step += self.state.batch_size * self.state.world_size
        self.state.loss_meter.update(loss.detach())

    def _run_loader(self, loader, steps=None):
        self.state.loss_meter.reset()
        for m in self.state.metric_meters.values():
            m.reset()
        self.state.epoch_size = steps or len(loader)
        m = _unwrap(self.state.model)
        if self.state.is_train and hasattr(m, "set_drop_position"):  # (epoch, step) -> generator position: a resumed run does not replay masks
            m.set_drop_position(self.state.epoch, 0)
        self.callbacks.on_loader_begin()
        with torch.set_grad_enabled(self.state.is_train):
            for i, batch in enumerate(loader):
                if i == self.state.epoch_size:
                    break
                self.state.step = i
                self.state.input = batch
                self.callbacks.on_batch_begin()
                self._make_step()
                self.callbacks.on_batch_end()
        self._reduce_meters()
        self.callbacks.on_loader_end()

    def _reduce_meters(self):
        """C5 of SURVEY §2.3: loss / metric averages are all-reduced across ranks ("metrics here are already reduced
        by runner", train.py:177)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        meters = [self.state.loss_meter] + list(self.state.metric_meters.values())
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([[m.sum, float(m.count)] for m in meters], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        for m, (s, c) in zip(meters, t.tolist()):
            m._sum, m._tsum, m.count = s, None, int(c)
