"""bres_fuse_check.py — BResNet-50 (BASELINE configs[3]) backward with bn1 / bn2's backward sums in the data gradients' epilogues (default) against the reduction
passes (MI355_BRESNET_FUSE_BN_BWD=0): per-segment relative difference of the gradients of one step (same forward)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sota_imagenet_amd.bresnet import BResNet50  # noqa: E402
from sota_imagenet_amd.synth import synthetic_batch  # noqa: E402
import oracle.resnet50_ref as R  # noqa: E402

N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 224
data, target = synthetic_batch(N, S, seed=0, index=3, device="cuda")
grads = []
for sw in ("1", "0"):
    os.environ["MI355_BRESNET_FUSE_BN_BWD"] = sw
    m = BResNet50(dtype="bf16", drop_rate=0.0, drop_connect_rate=0.2, weight_standardization=True, seed=4).cuda()
    m.train()
    R.smooth_ce(m(data), target, 0.1).backward()
    torch.cuda.synchronize()
    grads.append(m.flat_grads.detach().clone())
    segs = m._segments
errs = [((grads[0][b:e] - grads[1][b:e]).norm() / grads[1][b:e].norm().clamp_min(1e-30)).item() for b, e in segs]
print("per-segment relative difference:", ["%.1e" % x for x in errs])
print("max", max(errs), "equal" if torch.equal(grads[0], grads[1]) else "differ")
