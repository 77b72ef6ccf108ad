"""kernel_table.py [dtype] [N] [S] — one training step of ResNet-50 on the GPU, then the executor's record of which kernel every convolution's
forward / data-gradient / weight-gradient launch went to (mi355_resnet50_kernel_table).  `--write` also stores the table as
gpurun_out/kernel_table_bs<N>_<dtype>.txt (the GPU box only hands gpurun_out/ back); copying that file over the fixture
tests/golden/kernel_table_bs256_bf16.txt — which tests/test_resnet_gpu.py::test_baseline_batch_rule_selected_variants asserts — is the deliberate step
after a selection rule was changed on purpose."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sota_imagenet_amd.losses import CrossEntropyLoss  # noqa: E402
from sota_imagenet_amd.models import resnet50  # noqa: E402
from sota_imagenet_amd.synth import synthetic_batch  # noqa: E402


def table(dtype="bf16", N=256, S=224):
    m = resnet50(dtype=dtype).cuda()
    m.train()
    data, target = synthetic_batch(N, S, seed=0, index=0)
    crit = CrossEntropyLoss(smoothing=0.1).cuda()
    crit(m(data.cuda()), target.cuda()).backward()
    torch.cuda.synchronize()
    return m.kernel_table((N, S, S))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--write"]
    dtype = args[0] if args else "bf16"
    N, S = (int(args[1]) if len(args) > 1 else 256), (int(args[2]) if len(args) > 2 else 224)
    t = table(dtype, N, S)
    text = "".join("%s fwd=%s dgrad=%s wgrad=%s\n" % (k, v["fwd"], v["dgrad"], v["wgrad"]) for k, v in t.items())
    sys.stdout.write(text)
    if "--write" in sys.argv:
        out = os.path.join(ROOT, "gpurun_out", "kernel_table_bs%d_%s.txt" % (N, dtype))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        open(out, "w").write(text)
