"""dbg_bres.py — executor against per-op graph of BResNet-50 at 4 x 64 px, gradient by gradient (rel. L2): where the two arms part.  DBG_GG=1
compares two per-op graphs (run-to-run determinism).  MI355_TRACE_KERNELS=1 prints every conv launch, MI355_TRACE_SYNC=1 drains the device after each."""
import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["MI355_BRESNET_FUSED_ADD"] = "0"; os.environ["MI355_BRESNET_FUSED_ECA"] = "0"
from sota_imagenet_amd.bresnet import BResNet50, BResNet50Graph
from sota_imagenet_amd.synth import synthetic_batch
from oracle import ops_ref as R
import test_variant_gpu as T
dev = torch.device("cuda:0")
N, S = 4, 64
kw = dict(dtype="bf16", drop_rate=0.2, drop_connect_rate=0.2, weight_standardization=False)
m, g = (BResNet50Graph(**kw) if os.environ.get("DBG_GG") else BResNet50(**kw)), BResNet50Graph(**kw)
g.load_state_dict({k: v.detach().clone().contiguous() for k, v in m.state_dict().items()})
m, g = m.cuda(), g.cuda()
data, target = synthetic_batch(N, S, seed=0, index=2, device="cuda")
masks = T._masks(N, 3)
mk = {"dc": [None if k is None else k.to(dev) for k in masks["dc"]], "do": masks["do"].to(dev)}
m.masks, g.masks = mk, mk
m.train(); g.train()
print("== exec fwd", file=sys.stderr); om = m(data); torch.cuda.synchronize()
print("== graph fwd", file=sys.stderr); og = g(data); torch.cuda.synchronize()
print("logits max diff", (om - og).abs().max().item())
sm, sg = m.state_dict(), g.state_dict()
for k in sm:
    if "running" in k and not torch.equal(sm[k], sg[k]):
        print("FWD differs:", k, (sm[k] - sg[k]).abs().max().item()); 
print("== exec bwd", file=sys.stderr); R.smooth_ce(om, target, 0.1).backward(); torch.cuda.synchronize()
print("== graph bwd", file=sys.stderr); R.smooth_ce(og, target, 0.1).backward(); torch.cuda.synchronize()
gm, gg = dict(m.named_parameters()), dict(g.named_parameters())
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()
for k in gm:
    r = rel(gm[k].grad, gg[k].grad)
    if "weight" in k and ("conv" in k or "downsample.0" in k):
        print("%-40s %.3e" % (k, r))
