"""The torch-CPU oracle (oracle/ops_ref.py) against an independent fp64 restatement in plain loops (oracle/naive_ref.py),
at the shapes of the committed fixtures.  The reference holds no vectors for this path, so this is the second pin of the
oracle's semantics (padding / stride phase, variance conventions, momentum, first-maximum tie-breaking, smoothing).
Tolerance 2e-6 of the tensor's largest magnitude: ops_ref computes in fp32."""
import importlib.util
import os

import numpy as np
import torch

from oracle import naive_ref as N
from oracle import ops_ref as R
from sota_imagenet_amd.synth import uniform_tensor

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)


def close(name, got, ref, tol=2e-6):
    got = got.detach().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
    assert err <= tol, (name, err)


def test_conv_against_naive_loops():
    for case in MG.CONV_CASES + [("odd", 1, 5, 5, 64, 64, 3, 2), ("k1s1", 1, 3, 3, 64, 64, 1, 1)]:
        x, w, dy = MG.conv_inputs(case)
        s, pad = case[7], case[6] // 2
        close(case[0] + " y", R.conv2d_fwd(x, w, s, pad), N.conv2d_fwd(x.numpy(), w.numpy(), s, pad))
        dx, dw = R.conv2d_bwd(x, w, dy, s, pad)
        ndx, ndw = N.conv2d_bwd(x.numpy(), w.numpy(), dy.numpy(), s, pad)
        close(case[0] + " dx", dx, ndx)
        close(case[0] + " dw", dw, ndw)


def test_bn_against_naive_formulas():
    x = uniform_tensor((2, 6, 6, 64), 2.0, 111) + 0.3
    res = uniform_tensor((2, 6, 6, 64), 1.0, 112)
    g, b = uniform_tensor((64,), 0.5, 113) + 1.5, uniform_tensor((64,), 1.0, 114)
    rm, rv = uniform_tensor((64,), 1.0, 115), uniform_tensor((64,), 0.5, 116).abs() + 0.5
    dout = uniform_tensor((2, 6, 6, 64), 1.0, 117)
    for residual, relu in ((res, True), (None, True), (None, False)):
        got = R.bn_train(x, g, b, rm, rv, residual, relu, momentum=0.1)
        ref = N.bn_train(x.numpy(), g.numpy(), b.numpy(), rm.numpy(), rv.numpy(), None if residual is None else residual.numpy(), relu, momentum=0.1)
        for nm, a, r in zip(("out", "running_mean", "running_var", "mean", "invstd"), got, ref):
            close("bn " + nm, a, r, 5e-6)
        gb = R.bn_train_bwd(x, g, b, dout, residual, relu)
        rb = N.bn_train_bwd(x.numpy(), g.numpy(), b.numpy(), dout.numpy(), None if residual is None else residual.numpy(), relu)
        for nm, a, r in zip(("dx", "dgamma", "dbeta", "dres"), gb, rb):
            if r is not None:
                close("bn " + nm, a, r, 2e-5)
    # momentum convention (train.py:76 patches it to BN_MOM): 0.01 keeps 99 % of the running value
    got = R.bn_train(x, g, b, rm, rv, None, True, momentum=0.01)
    ref = N.bn_train(x.numpy(), g.numpy(), b.numpy(), rm.numpy(), rv.numpy(), None, True, momentum=0.01)
    close("bn running_var @0.01", got[2], ref[2], 5e-6)


def test_maxpool_against_naive_window_scan():
    xp = torch.relu(uniform_tensor((1, 8, 8, 64), 1.0, 121))  # many exact zeros: ties inside a window
    y, _ = R.maxpool(xp)
    ny, _ = N.maxpool(xp.numpy())
    close("maxpool y", y, ny, 0)
    dy = uniform_tensor(tuple(y.shape), 1.0, 122)
    close("maxpool dx", R.maxpool_bwd(xp, dy), N.maxpool_bwd(xp.numpy(), dy.numpy()), 1e-6)  # overlapping windows add: fp32 vs fp64 sums
    xo = uniform_tensor((2, 7, 9, 64), 1.0, 123)  # odd sizes: the padded border
    close("maxpool y odd", R.maxpool(xo)[0], N.maxpool(xo.numpy())[0], 0)


def test_ce_sgd_against_naive_formulas():
    logits = uniform_tensor((5, 1000), 6.0, 131)
    lab = torch.tensor([3, 999, 0, 512, 77])
    onehot = torch.nn.functional.one_hot(lab, 1000).float()
    soft = 0.7 * onehot + 0.3 * torch.nn.functional.one_hot((lab + 11) % 1000, 1000).float()
    for t in (onehot, soft):
        for s in (0.0, 0.1):
            l, dl = R.smooth_ce_bwd(logits, t, s)
            nl, ndl = N.smooth_ce(logits.numpy(), t.numpy(), s)
            assert abs(l.item() - nl) < 2e-6 * abs(nl)
            close("ce dlogits", dl, ndl, 5e-6)
    p0 = uniform_tensor((1003,), 1.0, 141)
    gs = [uniform_tensor((1003,), 1.0, 142 + i) for i in range(3)]
    p, m = R.sgd_steps(p0, gs, 0.1, 0.9, 3e-5)
    npp, nm = N.sgd_steps(p0.numpy(), [g.numpy() for g in gs], 0.1, 0.9, 3e-5)
    close("sgd p", p, npp)
    close("sgd m", m, nm)
