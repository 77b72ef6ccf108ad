"""Throughput of the reference-shaped loop (fit_wrapper.Runner + BatchMetrics + PhasesScheduler + ConsoleLogger-free) against the bare
step loop of bench.py: what the Runner's per-batch host work costs at batch 256, bf16.   python tools/runner_throughput.py [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sota_imagenet_amd import fit_wrapper as fw  # noqa: E402
from sota_imagenet_amd.data import SyntheticLoader  # noqa: E402
from sota_imagenet_amd.losses import CrossEntropyLoss  # noqa: E402
from sota_imagenet_amd.models import resnet50  # noqa: E402
from sota_imagenet_amd.optim import SGD  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = 256
model = resnet50(dtype="bf16").cuda()
crit = CrossEntropyLoss(smoothing=0.1).cuda()
opt = SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=3e-5)
loader = SyntheticLoader(dict(batch_size=N, image_size=224, num_classes=1000), size=N * steps, seed=0, device="cuda", pool=8)
cbs = [fw.BatchMetrics([fw.Accuracy(), fw.Accuracy(5)]), fw.PhasesScheduler([{"ep": [0, 1], "lr": [0.001, 0.1], "mode": "linear"}, {"ep": [1, 4], "lr": [0.1, 0.0], "mode": "cos"}])]
runner = fw.Runner(model, opt, crit, callbacks=cbs, use_fp16=False)
runner.fit(loader, steps_per_epoch=10, epochs=1)  # warm-up (context creation, pool generation)
torch.cuda.synchronize()
t0 = time.perf_counter()
runner.fit(loader, steps_per_epoch=steps, epochs=3, start_epoch=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = 2 * steps
print(f"Runner: {dt / n * 1e3:.3f} ms/step  {N * n / dt:.0f} img/s over {n} steps")
