"""per-segment gradient agreement of the fp8 step with the bf16 step (same parameters, same batch) — which direction costs what"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sota_imagenet_amd.losses import CrossEntropyLoss
from sota_imagenet_amd.models import resnet50
from sota_imagenet_amd.synth import synthetic_batch

N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 64
crit = CrossEntropyLoss(smoothing=0.1).cuda()
batches = [synthetic_batch(N, S, seed=5, index=i, device="cuda") for i in range(3)]

def run(dtype, env):
    for k, v in env.items():
        os.environ[k] = v
    m = resnet50(dtype=dtype).cuda(); m.train()
    out = []
    for d, t in batches:
        m.mark_grads_clean()
        l = crit(m(d), t); l.backward(); torch.cuda.synchronize()
        out.append((l.item(), m.flat_grads.clone()))
    for k in env:
        del os.environ[k]
    return out, m.grad_segments

ref, segs = run("bf16", {})
for name, env in [("fwd only", {"MI355_FP8_BWD": "0"}), ("bwd only", {"MI355_FP8_FWD": "0"}), ("both", {})]:
    got, _ = run("fp8", env)
    for i in (1, 2):
        cs = [torch.nn.functional.cosine_similarity(got[i][1][b:e], ref[i][1][b:e], dim=0).item() for b, e in segs]
        print(f"{name:9s} step {i} loss {got[i][0]:.4f} vs {ref[i][0]:.4f}  cos/segment:", " ".join(f"{c:.3f}" for c in cs))
