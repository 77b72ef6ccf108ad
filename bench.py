#!/usr/bin/env python3
"""bench.py — images/sec of the ResNet-50 training hot path on N MI355X (one process per GPU over RCCL).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch already resident in HBM: forward (conv+BN+ReLU blocks,
GAP, FC), label-smoothed CE, backward, gradient all-reduce (N > 1), SGD-momentum update — nothing skipped.
Workload = BASELINE.json metric config: ResNet-50, bs 256 per GPU, 224 px (configs[1] fp32 with --dtype fp32,
configs[2] bf16 by default), weak scaling.  Rank 0 prints ONE JSON line; besides the contract fields it carries
  roofline     — the dominant conv kernel (by HIP-event time inside the timed region, on the launch stream):
                 algorithmic FLOPs / measured time against the dense MFMA peak of the dtype
  cpu_baseline — the torch-CPU oracle (oracle/resnet50_ref.py) timed on this host's cores on a bounded sample
                 (BASELINE.json configs[0]: bs 32, fp32, 224 px), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2516.6, "fp8": 5033.2}  # MI355X_MICROARCH.md: dense MFMA peaks (no sparsity)
# what the matrix pipe SUSTAINS on random operands (the clock drops from 2.4 to ~2.0 GHz under a dense MFMA stream on real data): measured on this part with
# tools/micro/mfma_shape_random.hip, profiles/r06_mfma_shape_random_data.txt — reported beside `frac` (which stays against the datasheet peak), never instead of it
SUSTAINED_TFLOPS = {"bf16": 2016.0, "fp8": 4061.0}
FP8_KIND = 9  # profile_read kind: the conv launches that ran on e4m3 operands
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
EVENT_STEPS = 1  # timed steps whose conv / BN launches carry HIP-event pairs for `roofline` (each pair is a barrier on its stream: such a
                 # step is ~1.8 ms slower, and that is part of `value` — 1 step = 90 timed launches of the dominant class, 53 of the HBM kernel)
HBM_CLASS = 7  # executor profile class of bn_bwd_apply_kernel, the largest HBM-bound kernel of the step
# profile class -> kernel symbols (template instances of one source; names as tools/pmc_traffic.py writes them)
KERNEL_NAMES = {0: ["igemm_kernel<{T},128,128>", "igemm_kernel<{T},256,256>", "igemm_kernel<{T},256,128>", "igemm8_kernel<224,256>", "igemm8_kernel<256,128>",
                    "dconv_l2", "dconv_l3", "dconv_l4", "pw_k256_n1024", "pk_k1024_n256_w196", "pk_k2048_n512_w98", "pk_k512_n256_w196", "pk_k1024_n512_w196", "pk_k512_n2048_w196", "pk_k512_n128_w196",
                    "po_k64_b256", "po_k128_b256", "po_k256_b256", "po_k512_b128", "po_k256_b128"],  # (bf16 only: the generated assembly kernels of asm/dconv_gen.py, pw_gen.py, pk_gen.py, po_gen.py; class by Cout % 128 as the executor files them)
                1: ["igemm_kernel<{T},128,64>", "dconv_l1"],
                # (wg3_* / wg1_*: the generated kernels of asm/wg_gen.py / asm/wg1_gen.py, bf16 only; class by Cout % 128 as the executor files them)
                2: ["wgrad_kernel<{T},128,128>", "wgrad_kernel<{T},128,64>", "wg3_l2", "wg3_l3", "wg3_l4", "wg1_c1024_o256", "wg1_c256_o1024", "wg1_c2048_o512",
                    "wg1_c512_o2048", "wg1_c512_o256", "wg1_c1024_o512", "wg1_c512_o128", "wg1_c64_o256"],
                3: ["wgrad_kernel<{T},64,128>", "wgrad_kernel<{T},64,64>", "wg3_l1"]}


def runner_rate(N, S, steps):
    """img/s of fit_wrapper.Runner.fit (BatchMetrics with Acc@1 / Acc@5, PhasesScheduler) at batch N: the Runner path of train.py"""
    import time

    import torch

    from sota_imagenet_amd import fit_wrapper as fw
    from sota_imagenet_amd.data import SyntheticLoader
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD

    model = resnet50(dtype="bf16").cuda()
    crit = CrossEntropyLoss(smoothing=0.1).cuda()
    opt = SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=3e-5)
    loader = SyntheticLoader(dict(batch_size=N, image_size=S, num_classes=1000), size=N * steps, seed=0, device="cuda", pool=8)
    cbs = [fw.BatchMetrics([fw.Accuracy(), fw.Accuracy(5)]),
           fw.PhasesScheduler([{"ep": [0, 1], "lr": [0.001, 0.1], "mode": "linear"}, {"ep": [1, 4], "lr": [0.1, 0.0], "mode": "cos"}])]
    runner = fw.Runner(model, opt, crit, callbacks=cbs, use_fp16=False)
    runner.fit(loader, steps_per_epoch=8, epochs=1)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.fit(loader, steps_per_epoch=steps, epochs=2, start_epoch=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del runner, model
    torch.cuda.empty_cache()
    return {"workload": "fit_wrapper.Runner.fit: the same bf16 step behind the Runner loop (BatchMetrics Acc@1/Acc@5 + PhasesScheduler)",
            "value": round(N * steps / dt, 1), "unit": "images/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3)}


def pmc_traffic(kernels, dtype, batch, size):
    """HBM bytes per launch (launch-weighted over the symbols `kernels`) from the committed rocprofv3 --pmc passes of this
    same command (profiles/*_pmc_traffic_<dtype>.json, made by tools/pmc_traffic.py; PMC cannot be collected from inside
    the timed run).  Only valid for the default workload; None otherwise."""
    import glob

    if (batch, size) != (256, 224):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_{dtype}.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        table = json.load(f)["kernels"]
    recs = [table[k] for k in kernels if k in table]
    n = sum(r["launches"] for r in recs)
    pmc_traffic.source = "profiles/%s: separate --pmc passes of a serial step, not measured in this run" % os.path.basename(files[-1])
    return int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in recs) / n) if n else None


def step_traffic(dtype, batch, size):
    """HBM bytes of ONE whole train step: every kernel of the committed PMC passes (launches x corrected bytes per launch),
    divided by the steps those passes ran (= launches of the once-per-step SGD kernel).  None outside the default workload."""
    import glob

    if (batch, size) != (256, 224):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_{dtype}.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        table = json.load(f)["kernels"]
    steps = sum(r["launches"] for k, r in table.items() if k.startswith("sgd_kernel"))  # (sgd_kernel<false> since the EMA form exists)
    if not steps:
        return None, None
    return int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in table.values()) / steps), os.path.basename(files[-1])


def bresnet_step_hbm(batch, size, ms):
    """the BResNet-50 step (configs[3]) against the HBM roof: every kernel's counter traffic of one step (profiles/*_pmc_traffic_bresnet50_bf16.json, made by
    tools/profile_bres.sh -> tools/pmc_total.py from separate --pmc passes of this same command) over the step time.  None outside bs 256 / 224 px."""
    import glob

    if (batch, size) != (256, 224):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_bresnet50_bf16.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        prof = json.load(f)
    tb = prof["hbm_bytes_per_step"]
    gbs = tb / (ms * 1e-3) / 1e9
    # the bytes are a STORED profile's (a serial trace of the build named in the file name), only the step time is this run's: `derived` says so, and
    # the profile's kernel count travels with the number so that a reader can compare it with a fresh trace of the current build
    return {"bound": "hbm", "traffic": tb, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "derived": "traffic of a stored profile / step time of this run",
            "profile_kernels_in_step": prof.get("kernels_in_step"), "profile_commit": prof.get("commit"),
            "source": f"profiles/{os.path.basename(files[-1])}: every kernel of a serial step, separate --pmc passes"}


def cpu_baseline():
    """BASELINE.md §3 protocol: the oracle's train step (fwd + CE + bwd + SGD) and its forward alone on the host cores,
    BASELINE.json configs[0] (bs 32, fp32, 224 px): 3 warm-up + 10 timed steps each, median; threads printed."""
    import statistics

    from oracle import resnet50_ref as O
    from sota_imagenet_amd.synth import init_state_dict, synthetic_batch

    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))  # a one-GPU box owns a 16-core share of the host
    ref = O.ResNet50Ref()
    shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    ref.load_state_dict(init_state_dict(shapes, seed=0))
    O.patch_bn_mom(ref, 0.1)
    bs = 32
    batch = synthetic_batch(bs, 224, seed=0, index=0)
    opt = torch.optim.SGD(ref.parameters(), lr=0.001, momentum=0.9, weight_decay=3e-5)
    ref.train()

    def step():
        loss = O.smooth_ce(ref(batch[0]), batch[1], 0.1)
        opt.zero_grad()
        loss.backward()
        opt.step()

    def fwd():
        with torch.no_grad():
            O.smooth_ce(ref(batch[0]), batch[1], 0.1)

    def timed(fn, warm=3, n=10):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    t_train = timed(step)
    t_fwd = timed(fwd)
    return {"value": round(bs / t_train, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "forward_only_value": round(bs / t_fwd, 2),
            "sample": f"median of 10 (after 3 warm-up) fp32 train steps of bs={bs} @224px (BASELINE configs[0]) with the torch-CPU "
                      f"oracle; forward_only_value: the same for the forward + loss alone"}


def spawn_ranks(n):
    """the driver's own launch form (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>) as
    a child process; its stdout (rank 0's one JSON line) and stderr pass straight through.  Returns the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default=os.environ.get("BENCH_DTYPE", "bf16"), choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra fp32 (configs[1]) measurement")
    ap.add_argument("--model", default="resnet50", choices=["resnet50", "bresnet50"],
                    help="bresnet50: BASELINE configs[3] (variant graph, CutmixMixup on) — reported with its own workload name, no roofline")
    ap.add_argument("--spawn", action="store_true", help="launch the ranks from this process even for --gpus 1 (test hook)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # `python bench.py --gpus N` without a launcher: this process starts N fresh rank processes (one per GPU, RCCL between
        # them) BEFORE it touches HIP, waits for them and relays rank 0's JSON line.  Never an exec of a GPU-initialised process.
        raise SystemExit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_ddp = world > 1 or os.environ.get("BENCH_FORCE_DDP") == "1"  # the latter: 1-rank rehearsal of the RCCL path
    if use_ddp:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD
    from sota_imagenet_amd.synth import synthetic_batch

    N, S = args.batch, args.size

    def fence():
        if use_ddp:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def run(dtype, steps, warmup, want_roof, N=N, S=S, variant=None, extra_mask=0):
        """W untimed + exactly K timed training steps in `dtype`; returns (seconds, final loss, model)."""
        variant = args.model == "bresnet50" if variant is None else variant
        kw = dict(stem_type="deep", antialias=True, attn_type="eca", norm_layer="inplaceabn", norm_act="leaky_relu", drop_rate=0.2,
                  drop_connect_rate=0.2, weight_standardization=True) if variant else {}
        model = resnet50(dtype=dtype, **kw).cuda()
        criterion = CrossEntropyLoss(smoothing=0.1).cuda()
        # both executors keep their parameters in one flat array: the native SGD updates it in a handful of launches
        opt = SGD([{"params": list(model.parameters())}], lr=0.0, momentum=0.9, weight_decay=3e-5)
        opt.attach_model(model)
        mixer = ema_buf = None
        if variant:
            from sota_imagenet_amd.callbacks import CutmixMixup
            mixer = CutmixMixup(1.0, 0.2, prob=0.5)
            # the recipe's ModelEma (BResNet50_encoder.yaml:59 ema_decay 0.9999, train.py:111-112) IN the step: the parameter average inside the SGD
            # kernel (optim.SGD.attach_ema, what fit_wrapper.ModelEma does under the Runner), the BN buffers' average by one lerp
            ema_par, ema_buf = model.flat_params.detach().clone(), model._flat_buffers.detach().clone()
            opt.attach_ema(model.flat_params, ema_par, 0.9999)
        net = model
        if use_ddp:
            from sota_imagenet_amd.parallel import FlatBucketDDP

            net = FlatBucketDDP(model, device_ids=[local_rank])
        pool = [synthetic_batch(N, S, seed=0, stream=rank, index=i, device="cuda") for i in range(8 if N <= 256 else 4)]  # a pool of 8
        model.train()

        def step(i):
            data, target = pool[i % len(pool)]
            if mixer is not None:  # BASELINE configs[3]: "mixup/cutmix on" — on-device sampling + one mix kernel per batch
                data, target = mixer._dev(data, target, 1.0, 0.2, 0.5, 3)
            lr = 0.001 + 0.0001 * (i % 8)  # the scheduler writes a new LR every batch (train.py:131)
            for g in opt.param_groups:
                g["lr"] = lr
            loss = criterion(net(data), target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            if ema_buf is not None:
                ema_buf.lerp_(model._flat_buffers, 1.0 - 0.9999)
            return loss

        dom = None
        for i in range(warmup):
            if want_roof and i == warmup - 1:
                model.profile((N, S, S), 0b1111)  # last warm-up step: which conv kernel class is the dominant one
            step(i)
        if want_roof:
            tot = [model.profile_read((N, S, S), k)[0] for k in range(4)]
            dom = max(range(4), key=lambda k: tot[k])
            model.profile((N, S, S), 0)
        nprof = min(EVENT_STEPS, steps)  # a timed event pair drains the queue around its kernel: sample the last few timed steps only
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            if want_roof and i == steps - nprof:
                # HIP events around the dominant conv class and the largest HBM-bound kernel, inside the timed region
                model.profile((N, S, S), (1 << dom) | 0b11 | (1 << HBM_CLASS) | extra_mask)  # 0b11: both igemm classes (the 3x3 convs)
            loss = step(warmup + i)
        fence()
        dt = time.perf_counter() - t0
        if use_ddp:
            import torch.distributed as dist

            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, loss.item(), model, dom

    shape = (N, S, S)
    if args.model == "bresnet50":
        args.no_roofline = args.no_secondary = True
    want_roof = (not args.no_roofline) and rank == 0
    dt, final_loss, model, dom = run(args.dtype, args.steps, args.warmup, want_roof)

    def roof_of(model, dom, dtype, shape=shape):
        """(roofline, roofline_hbm) dicts from the HIP events recorded in the last timed steps of `model`"""
        roof = roof_hbm = None
        N = shape[0]
        # fp8: the launches that ran on e4m3 operands (igemm8_kernel<..., EB = 1> and the generated dconv_*_q 3x3 kernels), against the fp8 peak
        tot_ms, launches, flops, nbytes = model.profile_read(shape, FP8_KIND if dtype == "fp8" else dom)
        tdt = "float" if dtype == "fp32" else "__bf16"
        if launches:
            ach = flops / (tot_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[dtype]
            knames = ["igemm8_kernel<224,256,EB=1>", "igemm8_kernel<256,128,EB=1>", "dconv_l{2,3,4}_s{0,1,2}_q"] if dtype == "fp8" else \
                [n.format(T=tdt) for n in KERNEL_NAMES[dom] if not (dtype == "fp32" and n.startswith(("igemm8", "dconv_", "pw_", "pk_", "po_", "wg3_", "wg1_")))]
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": pmc_traffic(knames, dtype, N, S), "traffic_source": getattr(pmc_traffic, "source", None), "kernel": " + ".join(knames),
                    "launches": launches, "avg_launch_ms": round(tot_ms / launches, 4),
                    "alg_gflop_per_launch": round(flops / launches / 1e9, 3),
                    "alg_bytes_per_launch": int(nbytes / launches),
                    "event_steps": min(EVENT_STEPS, args.steps),
                    # the roofline MODEL's ceiling for this class: min(MFMA peak, arithmetic intensity x HBM peak).  `frac` stays
                    # achieved / MFMA peak (the number earlier rounds tracked); below the ridge the class is HBM-bound by the model
                    "intensity_flop_per_byte": round(flops / max(nbytes, 1), 1),
                    "ridge_flop_per_byte": round(peak * 1e12 / (PEAK_HBM_GBS * 1e9), 1),
                    "attainable": round(min(peak, flops / max(nbytes, 1) * PEAK_HBM_GBS * 1e9 / 1e12), 1),
                    "frac_of_attainable": round(ach / min(peak, flops / max(nbytes, 1) * PEAK_HBM_GBS * 1e9 / 1e12), 4),
                    "note": "timed with the weight-gradient side stream active: kernels of the two streams share the CUs, "
                            "so a launch takes longer than it does alone (serial_frac: same kernels, side stream off)"}
        t_ms, n_l, _, nbytes = model.profile_read(shape, HBM_CLASS)
        if n_l:
            gbs = nbytes / (t_ms * 1e-3) / 1e9
            roof_hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": pmc_traffic([f"bn_bwd_apply_kernel<{tdt}>"], dtype, N, S), "traffic_source": getattr(pmc_traffic, "source", None),
                        "kernel": f"bn_bwd_apply_kernel<{tdt}>", "launches": n_l, "avg_launch_ms": round(t_ms / n_l, 4),
                        "alg_bytes_per_launch": int(nbytes / n_l)}
        # BASELINE's conv target is quoted on the 3x3 convolutions: the same events, restricted to those launches
        t_ms, n_l, fl3, _ = model.profile_read(shape, 8)
        if n_l and roof is not None:
            roof["conv3x3"] = {"achieved": round(fl3 / (t_ms * 1e-3) / 1e12, 2), "frac": round(fl3 / (t_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 4),
                               "launches": n_l, "avg_launch_ms": round(t_ms / n_l, 4)}
        model.profile(shape, 0)
        return roof, roof_hbm

    def add_serial(roof, roof_class, dtype, steps):
        """the same kernel class with every kernel on one stream (MI355_WGRAD_STREAM=0), untimed extra steps"""
        os.environ["MI355_WGRAD_STREAM"] = "0"
        try:
            _, _, ms_, dom_s = run(dtype, steps, 2, True, extra_mask=0b1100)  # + the weight-gradient classes (untimed steps: their events cost nothing that is reported)
            t_ms, n_l, fl, _ = ms_.profile_read(shape, FP8_KIND if dtype == "fp8" else dom_s)
            t3, n3, fl3, _ = ms_.profile_read(shape, 8)
            tw = nw = fw_ = 0.0
            for cls in (2, 3):
                t_w, n_w, fl_w, _ = ms_.profile_read(shape, cls)
                tw, nw, fw_ = tw + t_w, nw + n_w, fw_ + fl_w
            if tw and dtype != "fp8":
                # the weight-gradient kernels (the largest symbol of the step): every kernel on one stream, like serial_frac
                tdt = "float" if dtype == "fp32" else "__bf16"
                roof["wgrad"] = {"serial_achieved": round(fw_ / (tw * 1e-3) / 1e12, 2), "serial_frac": round(fw_ / (tw * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 4),
                                 "launches": int(nw), "serial_avg_launch_ms": round(tw / nw, 4),
                                 "kernel": " + ".join(n.format(T=tdt) for c2 in (2, 3) for n in KERNEL_NAMES[c2])}
            ms_.profile(shape, 0)
            if n_l and (dom_s == roof_class or dtype == "fp8"):
                roof["serial_frac"] = round(fl / (t_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 4)
                roof["serial_avg_launch_ms"] = round(t_ms / n_l, 4)
            if n3 and "conv3x3" in roof:
                roof["conv3x3"]["serial_frac"] = round(fl3 / (t3 * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 4)
                if dtype in SUSTAINED_TFLOPS:
                    roof["conv3x3"]["sustained_peak_random_data"] = SUSTAINED_TFLOPS[dtype]
                    roof["conv3x3"]["serial_frac_of_sustained"] = round(fl3 / (t3 * 1e-3) / 1e12 / SUSTAINED_TFLOPS[dtype], 4)
            del ms_
        finally:
            del os.environ["MI355_WGRAD_STREAM"]
        torch.cuda.empty_cache()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * N * args.steps / dt
        roof = roof_hbm = None
        roof_class = dom
        if want_roof:
            roof, roof_hbm = roof_of(model, dom, args.dtype)
        if args.model == "bresnet50":
            print(json.dumps({"metric": "images/sec (whole node) BResNet-50 bs=%d/GPU @%dpx" % (N, S), "value": round(value, 1), "unit": "images/sec",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                              "config": {"workload": "BASELINE configs[3]: BResNet-50 (deep stem, anti-alias, ECA, leaky ABN, WS, drop-connect) train step "
                                                     "with CutmixMixup + ModelEma(0.9999) on, static executor (csrc/bresnet_exec.cpp)", "global_batch": world * N, "image_size": S,
                                         "parallelism": f"dp{world}", "final_loss": round(final_loss, 4)},
                              # whole step (every kernel, not one class): algorithmic conv + FC FLOPs of a training step / step time
                              "roofline": {"bound": "mfma", "scope": "whole step", "achieved": round(model.flops(N, S, S)[1] / (ms * 1e-3) / 1e12, 2),
                                           "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                           "frac": round(model.flops(N, S, S)[1] / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                                           "alg_gflop_per_step": round(model.flops(N, S, S)[1] / 1e9, 1)},
                              "step_hbm": bresnet_step_hbm(N, S, ms) if args.dtype == "bf16" else None,
                              "cpu_baseline": None}), flush=True)
            if use_ddp:
                import torch.distributed as dist

                dist.barrier()
                dist.destroy_process_group()
            return
        _, train_flops = model.flops(N, S, S)
        out = {
            "metric": "images/sec (whole node) ResNet-50 bs=256/GPU @224px" if args.dtype != "fp8" else f"images/sec (whole node) ResNet-50 bs={N}/GPU @{S}px, fp8 convs",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ResNet-50 v1.5 train step (fwd+CE+bwd+allreduce+SGD), bs={N}/GPU, {S}px, "
                                   + {"bf16": "configs[2] bf16 activations / fp32 accumulate+master", "fp32": "configs[1] fp32",
                                      "fp8": "configs[4] fp8 (e4m3) operands for the fwd / dgrad convs of layers 2-4, bf16 tensors + wgrad, delayed per-tensor scaling"}[args.dtype],
                       "global_batch": world * N, "image_size": S, "parallelism": f"dp{world}",
                       "step_tflops": round(train_flops / (dt / args.steps) / 1e12, 2), "final_loss": round(final_loss, 4)},
            "roofline": roof,
            "roofline_hbm": roof_hbm,
        }
        if world == 1:
            # the bf16 step as a whole is HBM-limited: all kernels' counter traffic of one step over the step time
            tb, src = step_traffic(args.dtype, N, S)
            if tb:
                gbs = tb / (ms * 1e-3) / 1e9
                out["step_hbm"] = {"bound": "hbm", "traffic": tb, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(gbs / PEAK_HBM_GBS, 4), "source": f"profiles/{src}: every kernel of a serial step, separate --pmc passes"}
        if roof is not None and world == 1 and not use_ddp:
            del model
            torch.cuda.empty_cache()
            add_serial(roof, roof_class, args.dtype, max(3, args.steps // 4))
            model = None
        if world == 1 and not use_ddp and args.dtype == "bf16" and not args.no_secondary:
            # BASELINE.json configs[1] (fp32, single MI355X) measured in the same process, with its own roofline
            model = None
            torch.cuda.empty_cache()
            k2 = max(4, args.steps // 4)
            dt2, loss2, m2, dom2 = run("fp32", k2, 2, want_roof)
            out["secondary"] = {"dtype": "fp32", "workload": "BASELINE configs[1]: ResNet-50 fp32 single MI355X bs=256 224px",
                                "value": round(N * k2 / dt2, 1), "unit": "images/sec", "steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 3),
                                "final_loss": round(loss2, 4)}
            if want_roof:
                r2, h2 = roof_of(m2, dom2, "fp32")
                del m2
                torch.cuda.empty_cache()
                if r2 is not None:
                    add_serial(r2, dom2, "fp32", 3)
                out["secondary"]["roofline"] = r2
                out["secondary"]["roofline_hbm"] = h2
            else:
                del m2
        if world == 1 and not use_ddp and args.dtype == "bf16" and not args.no_secondary:
            # BASELINE.json configs[4] on one GPU: bs 512, 224 px, fp8 (e4m3) convolutions — the step of §4.6 of DESIGN.md, with the
            # bf16 step at the SAME shape measured right beside it (same process, same box)
            model = None
            torch.cuda.empty_cache()
            try:
                k8 = max(8, args.steps // 2)
                dt8, loss8, m8, dom8 = run("fp8", k8, 3, want_roof, N=512, S=S)
                r8 = roof_of(m8, dom8, "fp8", shape=(512, S, S))[0] if want_roof else None  # the e4m3 conv launches against the fp8 MFMA peak
                del m8
                torch.cuda.empty_cache()
                dt16, loss16, m16, _ = run("bf16", k8, 3, False, N=512, S=S)
                del m16
                torch.cuda.empty_cache()
                out["secondary_fp8"] = {"dtype": "fp8", "workload": f"BASELINE configs[4] on one MI355X: ResNet-50 bs=512 {S}px, fp8 (e4m3) fwd / dgrad / wgrad "
                                                                    "operands for layers 2-4, delayed per-tensor scaling",
                                        "value": round(512 * k8 / dt8, 1), "unit": "images/sec", "steps": k8, "ms_per_step": round(dt8 / k8 * 1e3, 3),
                                        "final_loss": round(loss8, 4),
                                        "bf16_same_shape": {"value": round(512 * k8 / dt16, 1), "ms_per_step": round(dt16 / k8 * 1e3, 3), "final_loss": round(loss16, 4)}}
                if r8 is not None:
                    out["secondary_fp8"]["roofline"] = r8
            except Exception as e:  # the headline line must not depend on the extra measurement
                out["secondary_fp8"] = {"error": str(e)[:200]}
        if world == 1 and not use_ddp and args.dtype == "bf16" and not args.no_secondary:
            # BASELINE.json configs[3] on one GPU: the BResNet-50 variant graph with CutmixMixup on (bench.py --model bresnet50)
            model = None
            torch.cuda.empty_cache()
            try:
                kb = max(4, args.steps // 4)
                dtb, lossb, mb, _ = run("bf16", kb, 2, False, N=256, S=S, variant=True)
                flb = mb.flops(256, S, S)[1]
                del mb
                torch.cuda.empty_cache()
                out["secondary_bresnet50"] = {"dtype": "bf16", "workload": "BASELINE configs[3] on one MI355X: BResNet-50 (deep stem, anti-alias, ECA, leaky ABN, WS, "
                                                                           f"drop-connect) bs=256 {S}px, CutmixMixup + ModelEma(0.9999) on, static executor (csrc/bresnet_exec.cpp)",
                                              "value": round(256 * kb / dtb, 1), "unit": "images/sec", "steps": kb, "ms_per_step": round(dtb / kb * 1e3, 3),
                                              "final_loss": round(lossb, 4), "step_tflops": round(flb / (dtb / kb) / 1e12, 1),
                                              "step_hbm": bresnet_step_hbm(256, S, dtb / kb * 1e3)}
            except Exception as e:
                out["secondary_bresnet50"] = {"error": str(e)[:200]}
        if world == 1 and not use_ddp and args.dtype == "bf16" and not args.no_secondary and args.model != "bresnet50":
            # the reference's entry point is pytorch_tools' Runner (/root/reference/train.py:145-173): the same step driven by
            # fit_wrapper.Runner with BatchMetrics + PhasesScheduler, on the same batch pool — what the loop around the step costs
            model = None
            torch.cuda.empty_cache()
            try:
                out["secondary_runner"] = runner_rate(N, S, max(10, args.steps))
                out["secondary_runner"]["vs_value"] = round(out["secondary_runner"]["value"] / out["value"], 4)
            except Exception as e:
                out["secondary_runner"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_ddp:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
