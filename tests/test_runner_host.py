"""Host logic of the loop surface (no GPU): Runner / callbacks drive a tiny plain-torch model exactly as the reference's
Runner would (train.py:145-173); scheduler and metrics are pinned against the oracle restatements."""
import math
import os

import torch
import torch.nn as nn

from oracle import ops_ref as R
from oracle import resnet50_ref as O
from sota_imagenet_amd import fit_wrapper as fw


class TinyLoader:
    def __init__(self, n_batches=6, bs=8, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.data = [(torch.randn(bs, 10, generator=g), torch.nn.functional.one_hot(torch.randint(0, 4, (bs,), generator=g), 4).float())
                     for _ in range(n_batches)]
        self.batch_size = bs

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data)


def soft_ce(out, target):
    return -(torch.log_softmax(out, 1) * target).sum(1).mean()


def make_runner(callbacks, accumulate_steps=1):
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(10, 16), nn.ReLU(), nn.Linear(16, 4))
    opt = torch.optim.SGD(model.parameters(), lr=0.0, momentum=0.9)
    return fw.Runner(model, opt, soft_ce, callbacks=callbacks, accumulate_steps=accumulate_steps), model, opt


def test_phases_scheduler_matches_oracle_table():
    """fixture (iv) of SURVEY §8c: the r50 recipe's LR at every batch of a short epoch grid."""
    stages = [dict(ep=(0, 8), lr=(0.001, 1.0), mode="linear"), dict(ep=(8, 90), lr=(1.0, 0), mode="cos")]
    sched = fw.PhasesScheduler(stages)
    state = fw.RunnerState(optimizer=torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=0.0))
    sched.set_state(state)
    state.epoch_size = 5
    for epoch in (0, 3, 7, 8, 9, 50, 89):
        state.epoch = epoch
        sched.on_epoch_begin()
        for step in range(5):
            state.step = step
            sched.on_batch_begin()
            assert abs(state.optimizer.param_groups[0]["lr"] - O.phase_lr(stages, epoch, step, 5)) < 1e-12
    # endpoints of the recipe (configs/hydra_exp/1.r50_baseline.yaml:41-44)
    assert abs(O.phase_lr(stages, 0, 0, 5) - 0.001) < 1e-12 and abs(O.phase_lr(stages, 8, 0, 5) - 1.0) < 1e-12
    assert O.phase_lr(stages, 89, 4, 5) < 1e-3


def test_accuracy_matches_oracle():
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(64, 1000, generator=g)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 1000, (64,), generator=g), 1000).float()
    logits[torch.arange(0, 64, 3), tgt.argmax(1)[::3]] += 6.0
    for k in (1, 5):
        assert abs(fw.Accuracy(k)(logits, tgt).item() - R.accuracy(logits, tgt, k).item()) < 1e-5
    assert fw.Accuracy(5).name == "Acc@5"


def test_fit_loop_hooks_metrics_and_lr():
    calls = []

    class Spy(fw.Callback):
        def on_begin(self): calls.append("begin")
        def on_epoch_begin(self): calls.append(f"epoch{self.state.epoch}")
        def on_batch_begin(self): calls.append("b")
        def on_after_backward(self): calls.append("ab")
        def on_batch_end(self): calls.append("e")
        def on_end(self): calls.append("end")

    stages = [dict(ep=(0, 2), lr=(0.1, 0.0), mode="linear")]
    runner, model, opt = make_runner([fw.BatchMetrics([fw.Accuracy(), fw.Accuracy(2)]), fw.PhasesScheduler(stages), fw.Timer(), Spy()])
    loader = TinyLoader()
    w0 = model[0].weight.detach().clone()
    runner.fit(loader, val_loader=TinyLoader(3, seed=5), epochs=2)
    assert calls[0] == "begin" and calls[-1] == "end" and calls.count("ab") == 12  # backward only in training
    assert calls.count("b") == 12 + 6
    assert not torch.equal(w0, model[0].weight)
    assert set(runner.state.val_metrics) == {"Acc@1", "Acc@2"} and 0 <= runner.state.val_metrics["Acc@1"].avg <= 100
    assert math.isfinite(runner.state.train_loss.avg) and runner.state.global_sample_step == 2 * 6 * 8
    # last training batch of the run: pct = 11/12 of the way from 0.1 to 0
    assert abs(opt.param_groups[0]["lr"] - 0.1 * (1 - 11 / 12)) < 1e-12
    # steps_per_epoch (debug mode: train.py:168) truncates the epoch
    runner.fit(loader, steps_per_epoch=2, epochs=1)
    assert runner.state.epoch_size == 2


def test_accumulate_steps_and_evaluate():
    steps = []

    class CountingSGD(torch.optim.SGD):
        def step(self, *a, **k):
            steps.append(1)
            return super().step(*a, **k)

    torch.manual_seed(0)
    model = nn.Linear(10, 4)
    opt = CountingSGD(model.parameters(), lr=0.01)
    runner = fw.Runner(model, opt, soft_ce, callbacks=[fw.BatchMetrics(fw.Accuracy())], accumulate_steps=3)
    runner.fit(TinyLoader(6), epochs=1)
    assert len(steps) == 2
    loss, metrics = runner.evaluate(TinyLoader(2, seed=9))
    assert math.isfinite(loss) and len(metrics) == 1 and not model.training


def test_checkpoint_and_ema(tmp_path):
    runner, model, opt = make_runner([fw.BatchMetrics(fw.Accuracy()), fw.PhasesScheduler([dict(ep=(0, 1), lr=0.05)]),
                                      fw.CheckpointSaver(str(tmp_path), save_name="model.chpn", include_optimizer=True)])
    ema = fw.ModelEma(model, 0.5)
    runner.callbacks.callbacks.append(ema)
    ema.set_state(runner.state)
    runner.fit(TinyLoader(), val_loader=TinyLoader(2, seed=4), epochs=1)
    ck = torch.load(os.path.join(tmp_path, "model.chpn"))
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 1  # keys read back at train.py:98-109
    assert set(ck["state_dict"]) == set(model.state_dict())
    # after fit the live weights are back in the model and the EMA differs from them
    live = list(model.state_dict().values())
    assert any(not torch.equal(a, b) for a, b in zip(live, ema.ema))


def test_average_meter_defers_device_reads():
    m = fw.AverageMeter()
    m.update(torch.tensor(2.0))
    m.update(4.0)
    m.update(torch.tensor(6.0), n=2)
    assert abs(m.avg - (2 + 4 + 12) / 4) < 1e-9 and m.count == 4
