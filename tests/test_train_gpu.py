"""End-to-end drop-in surface on the GPU: train.py with a YAML config (Runner + callbacks + synthetic loader + native
model / loss / optimizer), evaluate-only + resume in the reference's checkpoint format, and the RCCL gradient path."""
import glob
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_py_runs_the_smoke_config(dev, tmp_path):
    sys.path.insert(0, ROOT)
    import train

    logdir = os.path.relpath(str(tmp_path), ROOT)
    val_loss, metrics = train.main(["+hydra_exp=test", f"log.dir={logdir}", "run.fp16=false", "random_seed=0",
                                    "data.pool=2", "log.save_optim=true"])
    assert val_loss == val_loss and 0.0 <= metrics["Acc@1"].avg <= 100.0 and metrics["Acc@5"].avg >= metrics["Acc@1"].avg
    run = glob.glob(os.path.join(str(tmp_path), "*_test", "*"))[0]
    logs = open(os.path.join(run, "logs.txt")).read()
    assert "Train loss:" in logs and "Acc@1:" in logs and "Model params: 25.56M" in logs  # line formats of the reference
    ck = torch.load(os.path.join(run, "model.chpn"), map_location="cpu")
    assert {"epoch", "state_dict", "optimizer"} <= set(ck) and "layer4.2.bn3.running_var" in ck["state_dict"]
    last = torch.load(os.path.join(run, "model_last.chpn"), map_location="cpu")
    assert last["conv1.weight"].shape == (64, 3, 7, 7)
    # evaluate-only from the checkpoint (train.py:158-162) in bf16 this time
    loss2, m2 = train.main(["+hydra_exp=test", f"log.dir={logdir}", f"run.resume={os.path.join(run, 'model.chpn')}",
                            "run.evaluate=true", "data.pool=2"])
    assert loss2 == loss2 and 0.0 <= m2["Acc@1"].avg <= 100.0


def test_train_py_runs_progressive_resize_with_fp8_convs(dev, tmp_path):
    """BASELINE configs[4] end to end at toy size: two stages with different image sizes (a new native ctx + its own fp8
    calibration step per shape), `model.dtype: fp8` from the YAML, through train.py / Runner / the stage manager."""
    sys.path.insert(0, ROOT)
    import train

    logdir = os.path.relpath(str(tmp_path), ROOT)
    val_loss, metrics = train.main(["+hydra_exp=progressive_fp8_test", f"log.dir={logdir}", "random_seed=0", "data.pool=2"])
    assert val_loss == val_loss and 0.0 <= metrics["Acc@1"].avg <= 100.0
    run = glob.glob(os.path.join(str(tmp_path), "*_progressive_fp8_test", "*"))[0]
    logs = open(os.path.join(run, "logs.txt")).read()
    assert logs.count("Train loss:") == 2 and "Model params: 25.56M" in logs
    losses = [float(x) for x in __import__("re").findall(r"Train loss: ([0-9.]+)", logs)]
    # (data.pool=2: the two synthetic batches repeat, so the fp8 step memorises them — the loss must FALL, from about ln 1000)
    assert len(losses) == 2 and losses[1] < losses[0] < 7.5 and losses[1] > 0.5, losses


def test_legacy_config_and_wd_filter_param_groups(dev, tmp_path):
    """legacy flat schema + `filter_from_wd` (train.py:83-86): BN / bias parameters land in a wd-0 group and the fused
    optimizer must not touch them with weight decay."""
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD

    sys.path.insert(0, ROOT)
    import train

    m = resnet50(dtype="fp32").cuda()
    groups = train.filter_from_weight_decay(m, ["bn", "bias"])
    assert len(groups[1]["params"]) == 53 * 2 + 1 and len(groups[0]["params"]) == 54
    opt = SGD(groups, lr=0.1, momentum=0.0, weight_decay=0.5)
    m.flat_grads.zero_()
    before = m.flat_params.clone()
    opt.step()
    p = dict(m.named_parameters())
    assert torch.equal(p["bn1.weight"], torch.ones_like(p["bn1.weight"]))  # wd 0 group, zero grad: untouched
    exp = 1.0 - 0.1 * 0.5
    ref = resnet50(dtype="fp32")
    assert torch.allclose(p["conv1.weight"].cpu(), dict(ref.named_parameters())["conv1.weight"].detach() * exp, rtol=1e-6, atol=0)
    # the padding rows behind fc.weight stay exactly zero
    assert torch.count_nonzero(m.flat_params[1024 + 1000 * 2048: 1024 + 1024 * 2048]) == 0


def test_bench_with_rccl_ddp_single_rank(dev):
    """the flat-bucket all-reduce path (side stream, events, ReduceOp.AVG over RCCL) with a 1-rank process group."""
    env = dict(os.environ, BENCH_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
                          "--warmup", "1", "--batch", "16", "--size", "64", "--dtype", "bf16", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["parallelism"] == "dp1"


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE): the parent spawns the ranks before touching HIP and
    relays rank 0's line — here N = 1 through the spawner (--spawn), the one-GPU rehearsal of the driver's 8-GPU command form."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--batch", "16",
                          "--size", "64", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["parallelism"] == "dp1"


def test_bench_bresnet50_line_with_rccl_single_rank(dev):
    """`bench.py --model bresnet50` (BASELINE configs[3] on its static executor) as a 1-rank process group: the flat gradient array goes
    through ONE native mean all-reduce behind the backward call; the line carries the whole-step roofline object."""
    env = dict(os.environ, BENCH_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29519")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
                          "--warmup", "1", "--batch", "8", "--size", "64", "--model", "bresnet50", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and "static executor" in rec["config"]["workload"]
    assert rec["roofline"]["scope"] == "whole step" and 0 < rec["roofline"]["frac"] < 1 and abs(rec["config"]["final_loss"] - 6.9) < 0.6


_DDP_CHECK = r"""
import os, sys, torch, torch.distributed as dist
from sota_imagenet_amd.losses import CrossEntropyLoss
from sota_imagenet_amd.models import resnet50
from sota_imagenet_amd.optim import SGD
from sota_imagenet_amd.parallel import FlatBucketDDP
from sota_imagenet_amd.synth import synthetic_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
crit = CrossEntropyLoss(smoothing=0.1)
batches = [synthetic_batch(4, 64, seed=31, stream=r, index=0, device="cuda") for r in range(world)]
# reference: every rank computes ALL ranks' gradients without DDP and averages them itself
ref = resnet50(dtype="fp32").cuda()
ref.train()
gsum = torch.zeros_like(ref.flat_grads)
for data, target in batches:
    ref.mark_grads_clean()
    crit(ref(data), target).backward()
    gsum += ref.flat_grads
gmean = gsum / world
# the DDP path: different initial parameters per rank (so the broadcast matters), own shard of the batch
m = resnet50(dtype="fp32").cuda()
with torch.no_grad():
    m.flat_params.mul_(1.0 + 0.1 * rank)
ddp = FlatBucketDDP(m, device_ids=[torch.cuda.current_device()], bucket_cap_mb=8.0)
assert torch.equal(m.flat_params, ref.flat_params), "rank-0 broadcast"
# what the communicator was actually asked to do (mi355_comm_stats): the broadcast covered parameters AND buffers
log = ddp.comm_stats()
assert log == [(1, 0, m.flat_params.numel()), (1, 0, m._flat_buffers.numel())], log
opt = SGD([{"params": list(m.parameters())}], lr=0.01, momentum=0.9, weight_decay=3e-5)
opt.attach_model(m)
m.train()
data, target = batches[rank]
loss = crit(ddp(data), target)
opt.zero_grad()
loss.backward()
torch.cuda.synchronize()
err = ((m.flat_grads - gmean).abs().max() / gmean.abs().max()).item()
assert err < 1e-6, f"all-reduced gradients vs the mean of the per-rank gradients: {err}"
# ONE backward = every element of the flat gradient array reduced exactly once, bucket after bucket, in plan order — true
# whatever the rank count (at 1 rank the numeric check above cannot see a wrong slice: ncclAvg is then the identity)
log = ddp.comm_stats()
plan = [(0, b, e) for b, e, _ in ddp.buckets]
assert len(plan) >= 3 and log == plan, (log, plan)
assert log[0][1] == 0 and log[-1][2] == m.flat_grads.numel() and all(a[2] == b[1] for a, b in zip(log, log[1:])), log
# no_sync (accumulate_steps > 1): micro-step 1 issues no collective and leaves the local gradient, micro-step 2 reduces the sum once
m.mark_grads_clean()
with ddp.no_sync():
    crit(ddp(data), target).backward()
assert ddp.comm_stats() == []
crit(ddp(data), target).backward()
assert ddp.comm_stats() == plan
torch.cuda.synchronize()
err2 = ((m.flat_grads - 2 * gmean).abs().max() / gmean.abs().max()).item()
assert err2 < 1e-5 * world, f"accumulated micro-steps, reduced once: {err2}"   # (same batch twice: local sum 2 g_r -> mean 2 gmean)
opt.step()
torch.cuda.synchronize()
mine = m.flat_params.clone()
other = mine.clone()
dist.broadcast(other, 0)
assert torch.equal(mine, other), "parameters identical across ranks after the step"
# a second step (the communicator's stream / events are reused; gradients overwrite)
loss = crit(ddp(data), target)
opt.zero_grad()
loss.backward()
opt.step()
torch.cuda.synchronize()
assert torch.isfinite(m.flat_params).all()
print(f"DDP-OK rank {rank} err {err:.2e} buckets {len(ddp.buckets)}")
dist.barrier()
dist.destroy_process_group()
"""


def _run_ddp_check(nranks, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    return subprocess.run(
                          [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr",
                           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_ddp_check.py")],
                          capture_output=True, text=True, env=env, timeout=900)


def test_native_rccl_allreduce_single_rank_numeric(dev, tmp_path):
    """the collective inside the C-ABI (csrc/comm.cpp) on a 1-rank RCCL communicator: rank-0 broadcast, the bucketed
    mean all-reduce chained behind backward on the communicator's stream, gradients == the mean (here: the gradient
    itself), parameters finite after two steps — the event / stream ordering of the real path, with numbers."""
    with open(os.path.join(ROOT, "tests", "_ddp_check.py"), "w") as f:
        f.write(_DDP_CHECK)
    try:
        out = _run_ddp_check(1, 29531)
    finally:
        os.remove(os.path.join(ROOT, "tests", "_ddp_check.py"))
    assert out.returncode == 0 and "DDP-OK rank 0" in out.stdout, (out.stdout[-800:], out.stderr[-2500:])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_native_rccl_allreduce_two_ranks_numeric(dev):
    """2 ranks, different batches and different initial parameters: all-reduced gradients == mean of the two single-GPU
    gradients (computed on each rank without DDP), parameters identical across ranks after the step."""
    with open(os.path.join(ROOT, "tests", "_ddp_check.py"), "w") as f:
        f.write(_DDP_CHECK)
    try:
        out = _run_ddp_check(2, 29533)
    finally:
        os.remove(os.path.join(ROOT, "tests", "_ddp_check.py"))
    assert out.returncode == 0 and "DDP-OK rank 0" in out.stdout and "DDP-OK rank 1" in out.stdout, (out.stdout[-800:], out.stderr[-2500:])


def test_train_py_runs_the_bresnet50_smoke_config(dev, tmp_path):
    """BASELINE configs[3] end to end at toy size: the variant model_params + weight standardisation + EMA + CutmixMixup
    through train.py / Runner, validation at a larger image size, checkpoint in the reference's format"""
    sys.path.insert(0, ROOT)
    import train

    logdir = os.path.relpath(str(tmp_path), ROOT)
    val_loss, metrics = train.main(["+hydra_exp=bresnet50_test", f"log.dir={logdir}", "run.fp16=true", "random_seed=0", "data.pool=2"])
    assert val_loss == val_loss and 0.0 <= metrics["Acc@1"].avg <= 100.0
    run = glob.glob(os.path.join(str(tmp_path), "*_bresnet50_test", "*"))[0]
    logs = open(os.path.join(run, "logs.txt")).read()
    assert "Train loss:" in logs and "Model params: 25.58M" in logs
    ck = torch.load(os.path.join(run, "model.chpn"), map_location="cpu")
    assert {"epoch", "state_dict"} <= set(ck) and "layer4.2.se_module.conv.weight" in ck["state_dict"] and "conv1.0.weight" in ck["state_dict"]


_DDP_VARIANT_CHECK = r"""
import os, torch, torch.distributed as dist
from sota_imagenet_amd.losses import CrossEntropyLoss
from sota_imagenet_amd.models import resnet50
from sota_imagenet_amd.parallel import FlatBucketDDP
from sota_imagenet_amd.synth import synthetic_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
kw = dict(stem_type="deep", antialias=True, attn_type="eca", norm_layer="inplaceabn", norm_act="leaky_relu", drop_rate=0.0, drop_connect_rate=0.0,
          weight_standardization=True, dtype="fp32")
crit = CrossEntropyLoss(smoothing=0.1)
batches = [synthetic_batch(2, 64, seed=41, stream=r, index=0, device="cuda") for r in range(world)]
ref = resnet50(**kw).cuda(); ref.train()
gsum = None
for data, target in batches:
    ref.zero_grad()
    crit(ref(data), target).backward()
    g = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    gsum = g.clone() if gsum is None else gsum + g
gmean = gsum / world
m = resnet50(**kw).cuda(); m.train()
with torch.no_grad():
    for p in m.parameters():
        p.mul_(1.0 + 0.1 * rank)
ddp = FlatBucketDDP(m, device_ids=[torch.cuda.current_device()], bucket_cap_mb=8.0)
assert all(torch.equal(a, b) for a, b in zip(m.parameters(), ref.parameters())), "rank-0 broadcast"
ddp.comm_stats()
data, target = batches[rank]
crit(ddp(data), target).backward()
torch.cuda.synchronize()
g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
err = ((g - gmean).abs().max() / gmean.abs().max()).item()
assert err < 1e-6, err
# the static executor reduces INSIDE its one backward call, bucket after bucket behind the segments that complete them: the
# communicator's own record is the plan, the buckets descend through the forward-ordered flat array and tile it exactly once
log = ddp.comm_stats()
plan = [(0, b, e) for b, e, _ in ddp.buckets]
assert len(plan) >= 3 and log == plan, (log, plan)
assert log[0][2] == m.flat_grads.numel() and log[-1][1] == 0 and all(a[1] == b[2] for a, b in zip(log, log[1:])), log
assert ddp.buckets == m.bucket_plan(8.0)
m.mark_grads_clean()
with ddp.no_sync():
    crit(ddp(data), target).backward()
assert ddp.comm_stats() == []
crit(ddp(data), target).backward()
assert ddp.comm_stats() == plan
torch.cuda.synchronize()
print(f"DDPV-OK rank {rank} err {err:.2e}")
dist.barrier(); dist.destroy_process_group()
"""


def test_native_rccl_allreduce_variant_model_single_rank(dev):
    """BResNet-50 on its static executor under FlatBucketDDP: bucketed mean all-reduces inside the ONE backward call
    (mi355_bresnet50_set_comm), checked against the mean of the per-rank gradients and against the communicator's own record
    (1-rank RCCL rehearsal with numbers; 2 ranks where available)"""
    path = os.path.join(ROOT, "tests", "_ddpv_check.py")
    with open(path, "w") as f:
        f.write(_DDP_VARIANT_CHECK)
    try:
        n = 2 if torch.cuda.device_count() >= 2 else 1
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                              "--master-port", "29541", path], capture_output=True, text=True, env=env, timeout=900)
    finally:
        os.remove(path)
    assert out.returncode == 0 and "DDPV-OK rank 0" in out.stdout, (out.stdout[-800:], out.stderr[-2500:])


def test_progressive_resize_and_val_batch_shapes(dev):
    """stage change (dali_dataloader.py:213-239): the same model runs at 64 px and 96 px and at another batch size; each
    shape gets its own native context, parameters stay shared."""
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.synth import synthetic_batch

    m = resnet50(dtype="bf16").cuda()
    crit = CrossEntropyLoss(smoothing=0.1)
    m.train()
    for N, S in [(4, 64), (4, 96), (6, 64)]:
        data, target = synthetic_batch(N, S, seed=4, index=S, device="cuda")
        loss = crit(m(data), target)
        m.mark_grads_clean()
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(m.flat_grads).all()
    assert len(m._ctxs) == 3
    m.eval()
    with torch.no_grad():
        out = m(synthetic_batch(5, 64, seed=4, index=9, device="cuda")[0])  # a 4th shape evicts the oldest context
    assert out.shape == (5, 1000) and len(m._ctxs) == 3


def test_runner_with_cutmix_mixup_soft_targets(dev):
    """BASELINE config 4 ingredient: CutMix/Mixup soft targets flow through the native CE and backward."""
    from sota_imagenet_amd import fit_wrapper as fw
    from sota_imagenet_amd.callbacks import CutmixMixup
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD
    from sota_imagenet_amd.synth import synthetic_batch

    class Loader:
        batch_size = 8

        def __len__(self):
            return 3

        def __iter__(self):
            return iter([synthetic_batch(8, 64, seed=5, index=i, device="cuda") for i in range(3)])

    m = resnet50(dtype="bf16").cuda()
    opt = SGD([{"params": list(m.parameters())}], lr=0.0, momentum=0.9, weight_decay=3e-5)
    runner = fw.Runner(m, opt, CrossEntropyLoss(smoothing=0.1),
                       callbacks=[fw.BatchMetrics([fw.Accuracy(), fw.Accuracy(5)]), fw.PhasesScheduler([dict(ep=(0, 1), lr=(0.001, 0.002))]),
                                  CutmixMixup(1.0, 0.2, prob=1.0)])
    before = m.flat_params.clone()
    runner.fit(Loader(), epochs=1)
    assert runner.state.train_loss.avg == runner.state.train_loss.avg  # finite
    assert not torch.equal(before, m.flat_params)


def test_model_ema_inside_the_sgd_kernel_matches_the_callback(dev):
    """ModelEma (train.py:111-112) over a flat-array model under the native SGD: the parameter average is advanced by the step kernel
    (optim.SGD.attach_ema).  The same three steps with the callback doing its own lerp (accumulate_steps = 2 keeps it unfused; here:
    forced by detaching) must leave the same average, and validation must run on the averaged weights and hand the live ones back."""
    from sota_imagenet_amd import fit_wrapper as fw
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD
    from sota_imagenet_amd.synth import synthetic_batch

    class Loader:
        batch_size = 4

        def __len__(self):
            return 3

        def __iter__(self):
            return iter([synthetic_batch(4, 64, seed=6, index=i, device="cuda") for i in range(3)])

    res = []
    for fused in (True, False):
        m = resnet50(dtype="fp32").cuda()
        opt = SGD([{"params": list(m.parameters())}], lr=0.0, momentum=0.9, weight_decay=3e-5)
        opt.attach_model(m)
        ema = fw.ModelEma(m, 0.9)
        if not fused:
            ema.on_begin = lambda: None  # the callback's own lerp after every batch
        runner = fw.Runner(m, opt, CrossEntropyLoss(smoothing=0.1), callbacks=[fw.PhasesScheduler([dict(ep=(0, 1), lr=(0.01, 0.02))]), ema])
        runner.fit(Loader(), val_loader=Loader(), epochs=1)
        assert ema._fused == fused and not ema._swapped
        res.append((m.flat_params.clone(), ema.ema[0].clone(), ema.ema[1].clone()))
    (p_a, e_a, b_a), (p_b, e_b, b_b) = res
    assert torch.equal(p_a, p_b) and torch.equal(b_a, b_b)
    assert not torch.equal(e_a, p_a)
    assert ((e_a - e_b).abs().max() / e_b.abs().max()).item() < 1e-6


def _tiny_steps(m, opt, idxs, N=4, S=64):
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.synth import synthetic_batch

    crit = CrossEntropyLoss(smoothing=0.1)
    m.train()
    for i in idxs:
        data, target = synthetic_batch(N, S, seed=21, index=i, device="cuda")
        loss = crit(m(data), target)
        opt.zero_grad()
        loss.backward()
        opt.step()


def test_sgd_resume_restores_momentum_bitwise(dev):
    """train.py:140-146 resume: after load_state_dict the next step must equal the uninterrupted run bit for bit (the
    loaded momentum buffers have to reach the flat momentum array the fused kernel reads)."""
    import copy

    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD

    def make():
        m = resnet50(dtype="fp32").cuda()
        opt = SGD([{"params": list(m.parameters())}], lr=0.05, momentum=0.9, weight_decay=3e-5)
        opt.attach_model(m)
        return m, opt

    m, opt = make()
    _tiny_steps(m, opt, [0, 1])
    ck = {"state_dict": copy.deepcopy(m.state_dict()), "optimizer": copy.deepcopy(opt.state_dict())}
    _tiny_steps(m, opt, [2])
    want = m.flat_params.clone()
    m2, opt2 = make()
    m2.load_state_dict(ck["state_dict"])
    opt2.load_state_dict(ck["optimizer"])
    _tiny_steps(m2, opt2, [2])
    assert torch.equal(m2.flat_params, want)
    # and without the optimizer state the step differs (the momentum term is not negligible at this lr)
    m3, opt3 = make()
    m3.load_state_dict(ck["state_dict"])
    _tiny_steps(m3, opt3, [2])
    assert not torch.equal(m3.flat_params, want)


def test_stock_torch_optimizer_and_module_zero_grad(dev):
    """the `_target_` schema allows any torch optimizer: grads set to None by model.zero_grad() / a stock optimizer's
    zero_grad(set_to_none=True) count as zeroed — the next backward must overwrite, not accumulate.  Checked exactly:
    after every backward the flat gradient equals the one a fresh model computes from the same parameters and batch."""
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.synth import synthetic_batch

    m = resnet50(dtype="fp32").cuda()
    fresh = resnet50(dtype="fp32").cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=3e-5)
    crit = CrossEntropyLoss(smoothing=0.1)
    m.train(), fresh.train()
    for i in range(4):
        data, target = synthetic_batch(4, 64, seed=22, index=i, device="cuda")
        loss = crit(m(data), target)
        if i % 2:
            m.zero_grad()  # nn.Module.zero_grad: set_to_none=True
        else:
            opt.zero_grad(set_to_none=True)
        loss.backward()
        with torch.no_grad():
            fresh.flat_params.copy_(m.flat_params)
        fresh.mark_grads_clean()
        crit(fresh(data), target).backward()
        assert torch.equal(m.flat_grads, fresh.flat_grads), f"step {i}: gradients accumulated instead of being overwritten"
        opt.step()
    assert torch.isfinite(m.flat_params).all()


def test_sgd_leaves_frozen_parameters_alone(dev):
    """train.py:44 filters requires_grad=False parameters out of the optimizer: their slice of the flat array sits
    between updated neighbours and must not be swept into a merged range."""
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD

    m = resnet50(dtype="fp32").cuda()
    frozen = dict(m.named_parameters())["layer3.2.bn2.weight"]
    frozen.requires_grad_(False)
    opt = SGD([{"params": [p for p in m.parameters() if p.requires_grad]}], lr=0.05, momentum=0.9, weight_decay=1e-2)
    opt.attach_model(m)
    before = frozen.detach().clone()
    other = dict(m.named_parameters())["layer3.2.bn2.bias"].detach().clone()
    _tiny_steps(m, opt, [0, 1])
    assert torch.equal(frozen.detach(), before)
    assert not torch.equal(dict(m.named_parameters())["layer3.2.bn2.bias"].detach(), other)
