"""reserve_cus_ab.py — what the bf16 step pays for CUs left to a collective, measured on ONE GPU.
  python tools/reserve_cus_ab.py <reserve> <standin workgroups>     one configuration per process (the reservation is process-global)
Every step issues 4 stand-in "all-reduces" (mi355_comm_standin: `workgroups` CU slots held for 500 us each — about what the four gradient
buckets of 44 / 38 / 17 / 3.5 MB take on a ring) on a separate high-priority stream during backward, the way comm.cpp issues the real ones."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
reserve, wgs = int(sys.argv[1]), int(sys.argv[2])
os.environ["MI355_RESERVE_CUS"] = str(reserve)
from sota_imagenet_amd import native  # noqa: E402
from sota_imagenet_amd.losses import CrossEntropyLoss  # noqa: E402
from sota_imagenet_amd.models import resnet50  # noqa: E402
from sota_imagenet_amd.optim import SGD  # noqa: E402
from sota_imagenet_amd.synth import synthetic_batch  # noqa: E402

L = native.lib()
N, S = 256, 224
m = resnet50(dtype="bf16").cuda()
crit = CrossEntropyLoss(smoothing=0.1).cuda()
opt = SGD([{"params": list(m.parameters())}], lr=0.0, momentum=0.9, weight_decay=3e-5)
opt.attach_model(m)
pool = [synthetic_batch(N, S, seed=0, index=i, device="cuda") for i in range(8)]
m.train()
side = torch.cuda.Stream(priority=-1)


def step(i):
    data, target = pool[i % 8]
    loss = crit(m(data), target)
    opt.zero_grad()
    if wgs:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):
                native.check(L.mi355_comm_standin(wgs, 500, side.cuda_stream))
    loss.backward()
    if wgs:
        torch.cuda.current_stream().wait_stream(side)
    opt.step()


for i in range(6):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for i in range(K):
    step(i)
torch.cuda.synchronize()
print(f"reserve {reserve:3d} CUs, stand-in {wgs:3d} workgroups x 4 x 500 us per step: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step", flush=True)
