"""Pins the oracle: oracle/*.py must reproduce the committed fixtures (tests/golden/*.npz, made by make_golden.py).
No reference test / golden vector exists for this path (parity unpinned by the reference), so the
fixtures are outputs of torch-CPU's own kernels; the only reference-given anchor is the parameter count "25.56M"
(configs/hydra_exp/1.r50_baseline.yaml:11), checked here as well."""
import importlib.util
import os

import numpy as np
import torch

from oracle import ops_ref as R
from oracle import resnet50_ref as O
from sota_imagenet_amd.synth import init_state_dict, synthetic_batch, uniform_tensor

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)
OPS = np.load(os.path.join(HERE, "golden", "ops_small.npz"))
NET = np.load(os.path.join(HERE, "golden", "resnet50_small.npz"))


def close(name, t, tol=2e-5, G=OPS):
    """compare a tensor with fixture `name` (sampled storage handled), normalised max error"""
    a = t.detach().numpy() if torch.is_tensor(t) else np.asarray(t)
    g = G[name]
    if name + "__abs_sum" in G.files:
        assert abs(np.abs(a.astype(np.float64)).sum() - G[name + "__abs_sum"]) <= tol * G[name + "__abs_sum"], name
        a = a.reshape(-1)[:: MG.STRIDE]
    assert a.shape == g.shape, (name, a.shape, g.shape)
    err = np.abs(a.astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-30)
    assert err <= tol, (name, err)


def test_param_count_anchor():
    assert int(NET["param_count"]) == 25_557_032 == sum(p.numel() for p in O.ResNet50Ref().parameters())


def test_conv_fixtures():
    for case in MG.CONV_CASES:
        x, w, dy = MG.conv_inputs(case)
        s, pad = case[7], case[6] // 2
        close(case[0] + "_y", R.conv2d_fwd(x, w, s, pad))
        dx, dw = R.conv2d_bwd(x, w, dy, s, pad)
        close(case[0] + "_dx", dx)
        close(case[0] + "_dw", dw)


def test_bn_pool_ce_sgd_accuracy_fixtures():
    x = uniform_tensor((2, 6, 6, 64), 2.0, 111) + 0.3
    res = uniform_tensor((2, 6, 6, 64), 1.0, 112)
    g, b = uniform_tensor((64,), 0.5, 113) + 1.5, uniform_tensor((64,), 1.0, 114)
    rm, rv = uniform_tensor((64,), 1.0, 115), uniform_tensor((64,), 0.5, 116).abs() + 0.5
    dout = uniform_tensor((2, 6, 6, 64), 1.0, 117)
    o, nrm, nrv, mean, invstd = R.bn_train(x, g, b, rm, rv, res, True)
    dx, dg, db, dres = R.bn_train_bwd(x, g, b, dout, res, True)
    for nm, t in dict(bn_out=o, bn_rm=nrm, bn_rv=nrv, bn_mean=mean, bn_invstd=invstd, bn_dx=dx, bn_dg=dg, bn_db=db, bn_dres=dres).items():
        close(nm, t, 1e-4)
    xp = torch.relu(uniform_tensor((1, 8, 8, 64), 1.0, 121))
    yp, _ = R.maxpool(xp)
    close("mp_y", yp, 0)
    close("mp_dx", R.maxpool_bwd(xp, uniform_tensor(tuple(yp.shape), 1.0, 122)), 0)
    logits = uniform_tensor((5, 1000), 6.0, 131)
    lab = torch.tensor([3, 999, 0, 512, 77])
    onehot = torch.nn.functional.one_hot(lab, 1000).float()
    soft = 0.7 * onehot + 0.3 * torch.nn.functional.one_hot((lab + 11) % 1000, 1000).float()
    for nm, t in (("hard", onehot), ("soft", soft)):
        for s in (0.0, 0.1):
            l, dl = R.smooth_ce_bwd(logits, t, s)
            assert abs(l.item() - float(OPS[f"ce_{nm}_{s}_loss"])) < 1e-5
            close(f"ce_{nm}_{s}_dl", dl)
    p0 = uniform_tensor((1003,), 1.0, 141)
    p, m = R.sgd_steps(p0, [uniform_tensor((1003,), 1.0, 142 + i) for i in range(3)], 0.1, 0.9, 3e-5)
    close("sgd_p", p, 1e-6)
    close("sgd_m", m, 1e-6)
    lg = uniform_tensor((32, 1000), 4.0, 151)
    tg = torch.nn.functional.one_hot((torch.arange(32) * 31) % 1000, 1000).float()
    lg[torch.arange(0, 32, 3), tg.argmax(1)[::3]] += 5.0
    assert abs(R.accuracy(lg, tg, 1).item() - float(OPS["acc1"])) < 1e-4 and abs(R.accuracy(lg, tg, 5).item() - float(OPS["acc5"])) < 1e-4


def test_resnet50_forward_fixture_64px():
    ref = O.ResNet50Ref()
    sd = init_state_dict([(k, tuple(v.shape)) for k, v in ref.state_dict().items()], seed=0)
    m = O.make_reference(sd)
    m.train()
    data, target = synthetic_batch(2, 64, seed=0, index=3)
    logits = m(data)
    loss = O.smooth_ce(logits, target, 0.1)
    loss.backward()
    close("logits_64", logits, 1e-3, NET)  # different host CPUs may sum in a different order
    assert abs(loss.item() - float(NET["loss_64"])) < 1e-4 * float(NET["loss_64"])
    close("bn1_running_var_64", m.bn1.running_var, 1e-4, NET)
    close("fc_weight_grad_64", m.fc.weight.grad[:8, :64], 1e-3, NET)


def test_lr_table_fixture():
    full = [dict(ep=(0, 8), lr=(0.001, 1.0), mode="linear"), dict(ep=(8, 90), lr=(1.0, 0), mode="cos")]
    tab = np.array([[O.phase_lr(full, e, s, 4) for s in range(4)] for e in range(90)])
    assert np.abs(tab - NET["lr_table"]).max() < 1e-12
