import sys, torch
sys.path.insert(0, '.')
from oracle import resnet50_ref as O
from sota_imagenet_amd.synth import synthetic_batch
from sota_imagenet_amd.models import resnet50
from sota_imagenet_amd.losses import CrossEntropyLoss

def nerr(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
def l2err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-20)).item()

for dtype in ["fp32", "bf16"]:
  for (N, S) in [(4, 64), (8, 128), (2, 224)]:
    m = resnet50(dtype=dtype)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ref = O.make_reference(sd)
    m = m.cuda()
    data, target = synthetic_batch(N, S, seed=0, index=3)
    ref.train(); out_ref = ref(data); loss_ref = O.smooth_ce(out_ref, target, 0.1); loss_ref.backward()
    m.train(); out = m(data.cuda()); loss = CrossEntropyLoss(smoothing=0.1)(out, target.cuda()); loss.backward()
    rp = dict(ref.named_parameters())
    errs = [(l2err(p.grad, rp[n].grad), nerr(p.grad, rp[n].grad), n) for n, p in m.named_parameters()]
    gflat = torch.cat([p.grad.detach().float().cpu().flatten() for n, p in m.named_parameters()])
    rflat = torch.cat([rp[n].grad.flatten() for n, p in m.named_parameters()])
    print(f"{dtype} N={N} S={S}: logits max {nerr(out, out_ref):.3e} l2 {l2err(out, out_ref):.3e} loss {loss.item():.6f} vs {loss_ref.item():.6f}"
          f" | grads: global l2 {l2err(gflat, rflat):.3e} worst-param l2 {max(errs)[0]:.3e} ({max(errs)[2]}) worst max-norm {max(e[1] for e in errs):.3e}")
    del m
    torch.cuda.empty_cache()
