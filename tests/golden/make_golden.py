#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the torch-CPU oracle (oracle/*.py).

The reference ships no tests / golden vectors for this path and its own Python cannot be imported here (its
dependencies — hydra, loguru, pytorch_tools, nvidia.dali — are not installed: ordinary ModuleNotFoundError, see
SURVEY.md §8c), so these vectors pin the ORACLE (torch 2.10 CPU kernels), not the reference: "parity unpinned".
Inputs are regenerated from seeds by sota_imagenet_amd.synth on both sides; only expected outputs are stored.

    python tests/golden/make_golden.py           # rewrites ops_small.npz, resnet50_small.npz and curve20.npz
    python tests/golden/make_golden.py curve20   # only the 20-step bs 32 / 224 px loss curve
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ops_ref as R  # noqa: E402
from oracle import resnet50_ref as O  # noqa: E402
from sota_imagenet_amd.synth import init_state_dict, synthetic_batch, uniform_tensor  # noqa: E402

# (name, N, H, W, Cin, Cout, K, stride)
CONV_CASES = [("c1x1", 2, 8, 8, 64, 64, 1, 1), ("c3x3", 2, 8, 8, 64, 128, 3, 1), ("c3x3s2", 2, 8, 8, 128, 64, 3, 2), ("c1x1s2", 1, 8, 8, 64, 128, 1, 2)]
CURVE = dict(N=8, S=64, steps=6, lr=(0.0005, 0.004))
# 20 SGD steps, bs 32, 224 px, seed 0, the linear-warm-up + cosine SHAPE of the r50 recipe
# (/root/reference/configs/hydra_exp/1.r50_baseline.yaml:41-44: lr [0.001, 1.0] linear over epochs 0-8, [1.0, 0] cos over 8-90)
# compressed to 5 "epochs" of 4 steps, peak scaled by the linear rule 1.0 * 32 / 1024 = 0.031
CURVE20 = dict(N=32, S=224, steps=20, epoch_size=4, peak=0.031,
               stages=[dict(ep=(0, 2), lr=(0.001 * 0.031, 0.031), mode="linear"), dict(ep=(2, 5), lr=(0.031, 0.0), mode="cos")])


def conv_inputs(case):
    name, N, H, W, Cin, Cout, K, s = case
    x = uniform_tensor((N, H, W, Cin), 1.0, 101)
    w = uniform_tensor((Cout, K, K, Cin), 0.1, 102)
    Ho = (H + 2 * (K // 2) - K) // s + 1
    dy = uniform_tensor((N, Ho, Ho, Cout), 1.0, 103)
    return x, w, dy


STRIDE = 7  # large arrays are stored as every 7th element + the sum of magnitudes (keeps the fixtures small)


def pack(out, name, t):
    a = t.detach().numpy() if torch.is_tensor(t) else np.asarray(t)
    if a.size > 4096:
        out[name] = a.reshape(-1)[::STRIDE].astype(np.float32)
        out[name + "__abs_sum"] = np.float64(np.abs(a.astype(np.float64)).sum())
    else:
        out[name] = a


def make_ops():
    out = {}
    for case in CONV_CASES:
        x, w, dy = conv_inputs(case)
        s, pad = case[7], case[6] // 2
        y = R.conv2d_fwd(x, w, s, pad)
        dx, dw = R.conv2d_bwd(x, w, dy, s, pad)
        pack(out, case[0] + "_y", y), pack(out, case[0] + "_dx", dx), pack(out, case[0] + "_dw", dw)
    # BN (+residual, +ReLU) train fwd/bwd incl. running-stat update
    x = uniform_tensor((2, 6, 6, 64), 2.0, 111) + 0.3
    res = uniform_tensor((2, 6, 6, 64), 1.0, 112)
    g, b = uniform_tensor((64,), 0.5, 113) + 1.5, uniform_tensor((64,), 1.0, 114)
    rm, rv = uniform_tensor((64,), 1.0, 115), uniform_tensor((64,), 0.5, 116).abs() + 0.5
    dout = uniform_tensor((2, 6, 6, 64), 1.0, 117)
    o, nrm, nrv, mean, invstd = R.bn_train(x, g, b, rm, rv, res, True)
    dx, dg, db, dres = R.bn_train_bwd(x, g, b, dout, res, True)
    for nm, t in dict(bn_out=o, bn_rm=nrm, bn_rv=nrv, bn_mean=mean, bn_invstd=invstd, bn_dx=dx, bn_dg=dg, bn_db=db, bn_dres=dres).items():
        pack(out, nm, t)
    # maxpool with ties, GAP
    xp = torch.relu(uniform_tensor((1, 8, 8, 64), 1.0, 121))
    yp, _ = R.maxpool(xp)
    pack(out, "mp_y", yp), pack(out, "mp_dx", R.maxpool_bwd(xp, uniform_tensor(tuple(yp.shape), 1.0, 122)))
    # CE: one-hot and soft targets, smoothing 0 and 0.1
    logits = uniform_tensor((5, 1000), 6.0, 131)
    lab = torch.tensor([3, 999, 0, 512, 77])
    onehot = torch.nn.functional.one_hot(lab, 1000).float()
    soft = 0.7 * onehot + 0.3 * torch.nn.functional.one_hot((lab + 11) % 1000, 1000).float()
    for nm, t in (("hard", onehot), ("soft", soft)):
        for s in (0.0, 0.1):
            l, dl = R.smooth_ce_bwd(logits, t, s)
            out[f"ce_{nm}_{s}_loss"] = np.float32(l.item())
            pack(out, f"ce_{nm}_{s}_dl", dl)
    # SGD: 3 steps incl. the step-0 momentum rule
    p0 = uniform_tensor((1003,), 1.0, 141)
    grads = [uniform_tensor((1003,), 1.0, 142 + i) for i in range(3)]
    p, m = R.sgd_steps(p0, grads, 0.1, 0.9, 3e-5)
    out.update(sgd_p=p.numpy(), sgd_m=m.numpy())
    # accuracy
    lg = uniform_tensor((32, 1000), 4.0, 151)
    tg = torch.nn.functional.one_hot((torch.arange(32) * 31) % 1000, 1000).float()
    lg[torch.arange(0, 32, 3), tg.argmax(1)[::3]] += 5.0
    out.update(acc1=np.float32(R.accuracy(lg, tg, 1).item()), acc5=np.float32(R.accuracy(lg, tg, 5).item()))
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **out)


def make_net():
    ref = O.ResNet50Ref()
    shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    sd = init_state_dict(shapes, seed=0)
    out = {"param_count": np.int64(sum(p.numel() for p in ref.parameters()))}
    # forward + backward, bs 2 @ 64 px and @ 224 px
    for S in (64, 224):
        m = O.make_reference(sd)
        m.train()
        data, target = synthetic_batch(2, S, seed=0, index=3)
        logits = m(data)
        loss = O.smooth_ce(logits, target, 0.1)
        loss.backward()
        out[f"logits_{S}"] = logits.detach().numpy()
        out[f"loss_{S}"] = np.float32(loss.item())
        out[f"fc_weight_grad_{S}"] = m.fc.weight.grad[:8, :64].numpy().copy()
        out[f"bn1_running_var_{S}"] = m.bn1.running_var.numpy().copy()
    # loss curves (fp32 and fp64 oracle) for the same seeds the GPU test uses
    N, S, steps = CURVE["N"], CURVE["S"], CURVE["steps"]
    batches = [synthetic_batch(N, S, seed=0, index=i) for i in range(steps)]
    stages = [dict(ep=(0, 1), lr=CURVE["lr"], mode="linear")]
    lrs = [O.phase_lr(stages, 0, i, steps) for i in range(steps)]
    l32, _ = O.train_steps(O.make_reference(sd), batches, lrs)
    l64, _ = O.train_steps(O.make_reference(sd).double(), [(d.double(), t.double()) for d, t in batches], lrs)
    out.update(curve_fp32=np.array(l32), curve_fp64=np.array(l64), curve_lrs=np.array(lrs))
    # LR table of the full recipe (fixture iv): epochs x 4 steps
    full = [dict(ep=(0, 8), lr=(0.001, 1.0), mode="linear"), dict(ep=(8, 90), lr=(1.0, 0), mode="cos")]
    out["lr_table"] = np.array([[O.phase_lr(full, e, s, 4) for s in range(4)] for e in range(90)])
    np.savez_compressed(os.path.join(HERE, "resnet50_small.npz"), **out)


def make_curve20():
    """fixture (iii) at the size SURVEY 8(c) specifies; fp32 and fp64 runs of the same oracle (the gap between them is the
    yardstick the GPU test scales its band with).  ~5 min on 8 host threads."""
    torch.use_deterministic_algorithms(True)
    torch.set_num_threads(8)
    ref = O.ResNet50Ref()
    sd = init_state_dict([(k, tuple(v.shape)) for k, v in ref.state_dict().items()], seed=0)
    c = CURVE20
    batches = [synthetic_batch(c["N"], c["S"], seed=0, index=i) for i in range(c["steps"])]
    lrs = [O.phase_lr(c["stages"], i // c["epoch_size"], i % c["epoch_size"], c["epoch_size"]) for i in range(c["steps"])]
    l32, _ = O.train_steps(O.make_reference(sd), batches, lrs)
    l64, _ = O.train_steps(O.make_reference(sd).double(), [(d.double(), t.double()) for d, t in batches], lrs)
    np.savez_compressed(os.path.join(HERE, "curve20.npz"), curve_fp32=np.array(l32), curve_fp64=np.array(l64), curve_lrs=np.array(lrs))
    print("curve20 fp32", np.round(l32, 4), "\n        fp64", np.round(l64, 4))


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1:] == ["curve20"]:
        make_curve20()
        sys.exit(0)
    make_ops()
    make_net()
    make_curve20()
    for f in ("ops_small.npz", "resnet50_small.npz", "curve20.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
