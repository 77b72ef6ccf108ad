import sys, torch
sys.path.insert(0, '.')
from oracle import resnet50_ref as O
from sota_imagenet_amd.synth import synthetic_batch
from sota_imagenet_amd.models import resnet50
from sota_imagenet_amd.losses import CrossEntropyLoss
def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()
def gflat(params): return torch.cat([p.grad.detach().double().cpu().flatten() for p in params])
def ce64(o, t):
    lp = torch.log_softmax(o.double(), 1); return ((0.9) * -(lp * t.double()).sum(1) + 0.1 * -lp.mean(1)).mean()
for (N, S) in [(8, 64), (8, 128), (2, 224)]:
    data, target = synthetic_batch(N, S, seed=0, index=3)
    res = {}
    for dtype in ["fp32", "bf16"]:
        m = resnet50(dtype=dtype)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        m = m.cuda(); m.train()
        out = m(data.cuda()); loss = CrossEntropyLoss(smoothing=0.1)(out, target.cuda()); loss.backward()
        res[dtype] = (out.detach().cpu(), loss.item(), gflat(m.parameters()))
        del m
    r64 = O.make_reference(sd).double(); r64.train(); o64 = r64(data.double()); l64 = ce64(o64, target); l64.backward(); g64 = gflat(r64.parameters())
    r32 = O.make_reference(sd); r32.train(); o32 = r32(data); l32 = O.smooth_ce(o32, target, 0.1); l32.backward(); g32 = gflat(r32.parameters())
    rb = O.make_reference(sd); rb.train()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ob = rb(data); 
    lb = O.smooth_ce(ob.float(), target, 0.1); lb.backward(); gb = gflat(rb.parameters())
    print(f"N={N} S={S}  (all vs fp64 oracle)  logits-l2 / loss / grad-l2")
    print(f"   torch-cpu fp32 : {l2(o32,o64):.2e}  {l32.item():.5f}  {l2(g32,g64):.2e}")
    print(f"   native    fp32 : {l2(res['fp32'][0],o64):.2e}  {res['fp32'][1]:.5f}  {l2(res['fp32'][2],g64):.2e}")
    print(f"   torch-cpu bf16 : {l2(ob,o64):.2e}  {lb.item():.5f}  {l2(gb,g64):.2e}")
    print(f"   native    bf16 : {l2(res['bf16'][0],o64):.2e}  {res['bf16'][1]:.5f}  {l2(res['bf16'][2],g64):.2e}   (loss64 {l64.item():.5f})")
