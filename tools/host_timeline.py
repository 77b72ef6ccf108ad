"""Host-side time of each call of the train step WITHOUT synchronising (how far ahead of the GPU the host runs, and which call — if
any — blocks until the GPU catches up).   python tools/host_timeline.py [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sota_imagenet_amd.losses import CrossEntropyLoss  # noqa: E402
from sota_imagenet_amd.models import resnet50  # noqa: E402
from sota_imagenet_amd.optim import SGD  # noqa: E402
from sota_imagenet_amd.synth import synthetic_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# N S from the environment: at a small batch (HOST_N=16) the GPU outruns the host, the launch queue never fills and the host loop time
# IS the unblocked enqueue cost of a step; at 256 x 224 some calls block on a full queue
import os
N, S = int(os.environ.get("HOST_N", 256)), int(os.environ.get("HOST_S", 224))
model = resnet50(dtype="bf16").cuda()
crit = CrossEntropyLoss(smoothing=0.1).cuda()
opt = SGD([{"params": list(model.parameters())}], lr=0.001, momentum=0.9, weight_decay=3e-5)
opt.attach_model(model)
pool = [synthetic_batch(N, S, seed=0, index=i, device="cuda") for i in range(4)]
model.train()
acc = {"fwd": 0.0, "ce": 0.0, "zero": 0.0, "bwd": 0.0, "opt": 0.0}
for i in range(steps + 5):
    if i == 5:
        torch.cuda.synchronize()
        acc = {k: 0.0 for k in acc}
        t_all = time.perf_counter()
    data, target = pool[i % 4]
    t0 = time.perf_counter(); out = model(data)
    t1 = time.perf_counter(); loss = crit(out, target)
    t2 = time.perf_counter(); opt.zero_grad()
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k] += v
host = time.perf_counter() - t_all
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print("host ms per call:", {k: round(v / steps * 1e3, 3) for k, v in acc.items()}, f"| host loop {host / steps * 1e3:.3f} ms/step, with final sync {total / steps * 1e3:.3f} ms/step")

if len(sys.argv) > 2 and sys.argv[2] == "profile":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(20):
        data, target = pool[i % 4]
        loss = crit(model(data), target)
        opt.zero_grad()
        loss.backward()
        opt.step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
