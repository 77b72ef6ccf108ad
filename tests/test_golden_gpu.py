"""The native path against the COMMITTED fixtures (tests/golden/*.npz) — no oracle code runs in these tests."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from sota_imagenet_amd.synth import synthetic_batch, uniform_tensor

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden_consts", os.path.join(HERE, "golden", "make_golden.py"))
OPS = np.load(os.path.join(HERE, "golden", "ops_small.npz"))
NET = np.load(os.path.join(HERE, "golden", "resnet50_small.npz"))
STRIDE = 7
CONV_CASES = [("c1x1", 2, 8, 8, 64, 64, 1, 1), ("c3x3", 2, 8, 8, 64, 128, 3, 1), ("c3x3s2", 2, 8, 8, 128, 64, 3, 2), ("c1x1s2", 1, 8, 8, 64, 128, 1, 2)]


def close(name, t, tol, G=OPS):
    a = t.detach().float().cpu().numpy()
    g = G[name]
    if name + "__abs_sum" in G.files:
        a = a.reshape(-1)[::STRIDE]
    assert a.shape == g.shape, (name, a.shape, g.shape)
    err = np.abs(a.astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-30)
    assert err <= tol, (name, err)


def test_conv_golden(dev):
    from sota_imagenet_amd import ops

    for name, N, H, W, Cin, Cout, K, s in CONV_CASES:
        x = uniform_tensor((N, H, W, Cin), 1.0, 101).to(dev)
        w = uniform_tensor((Cout, K, K, Cin), 0.1, 102).to(dev)
        Ho = (H + 2 * (K // 2) - K) // s + 1
        dy = uniform_tensor((N, Ho, Ho, Cout), 1.0, 103).to(dev)
        close(name + "_y", ops.conv2d_fwd(x, w, s, K // 2), 2e-5)
        close(name + "_dx", ops.conv2d_dgrad(dy, w, (N, H, W, Cin), s, K // 2), 2e-5)
        close(name + "_dw", ops.conv2d_wgrad(dy, x, K, K, s, K // 2), 2e-5)


def test_ce_sgd_golden(dev):
    from sota_imagenet_amd import ops

    logits = uniform_tensor((5, 1000), 6.0, 131).to(dev)
    lab = torch.tensor([3, 999, 0, 512, 77])
    onehot = torch.nn.functional.one_hot(lab, 1000).float()
    soft = 0.7 * onehot + 0.3 * torch.nn.functional.one_hot((lab + 11) % 1000, 1000).float()
    for nm, t in (("hard", onehot), ("soft", soft)):
        for s in (0.0, 0.1):
            loss, dl = ops.ce_loss(logits, t.to(dev), s)
            assert abs(loss.item() - float(OPS[f"ce_{nm}_{s}_loss"])) < 1e-5
            close(f"ce_{nm}_{s}_dl", dl, 1e-5)
    p = uniform_tensor((1003,), 1.0, 141).to(dev)
    m = torch.zeros_like(p)
    for i in range(3):
        ops.sgd_step(p, uniform_tensor((1003,), 1.0, 142 + i).to(dev), m, 0.1, 0.9, 3e-5)
    close("sgd_p", p, 1e-6)
    close("sgd_m", m, 1e-6)


def test_bn_pool_accuracy_golden(dev):
    """the committed bn_* / mp_* / acc* vectors (same inputs as tests/golden/make_golden.py) against the HIP kernels"""
    from sota_imagenet_amd import ops
    from sota_imagenet_amd.fit_wrapper import Accuracy

    x = (uniform_tensor((2, 6, 6, 64), 2.0, 111) + 0.3).to(dev)
    res = uniform_tensor((2, 6, 6, 64), 1.0, 112).to(dev)
    g, b = (uniform_tensor((64,), 0.5, 113) + 1.5).to(dev), uniform_tensor((64,), 1.0, 114).to(dev)
    rm, rv = uniform_tensor((64,), 1.0, 115).to(dev), (uniform_tensor((64,), 0.5, 116).abs() + 0.5).to(dev)
    dout = uniform_tensor((2, 6, 6, 64), 1.0, 117).to(dev)
    out, mean, invstd = ops.bn_fwd_train(x, g, b, rm, rv, residual=res, relu=True)
    close("bn_out", out, 1e-5)
    close("bn_rm", rm, 1e-5)
    close("bn_rv", rv, 1e-5)
    close("bn_mean", mean, 1e-5)
    close("bn_invstd", invstd, 1e-5)
    dx, dg, db, dz = ops.bn_bwd(dout, out, x, g, mean, invstd, relu=True, want_dz=True)
    close("bn_dx", dx, 1e-4)
    close("bn_dg", dg, 1e-4)
    close("bn_db", db, 1e-4)
    close("bn_dres", dz, 0)  # the residual branch's gradient is the masked incoming gradient: exact
    xp = torch.relu(uniform_tensor((1, 8, 8, 64), 1.0, 121)).to(dev)
    yp, idx = ops.maxpool_fwd(xp)
    close("mp_y", yp, 0)
    close("mp_dx", ops.maxpool_bwd(uniform_tensor(tuple(yp.shape), 1.0, 122).to(dev), idx, tuple(xp.shape)), 0)
    lg = uniform_tensor((32, 1000), 4.0, 151)
    tg = torch.nn.functional.one_hot((torch.arange(32) * 31) % 1000, 1000).float()
    lg[torch.arange(0, 32, 3), tg.argmax(1)[::3]] += 5.0
    for k, name in ((1, "acc1"), (5, "acc5")):
        acc = Accuracy(k)(lg.to(dev), tg.to(dev))
        assert abs(float(acc) - float(OPS[name])) < 1e-4, name


@pytest.mark.parametrize("S", [64, 224])
def test_resnet50_logits_golden(dev, S):
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50

    m = resnet50(dtype="fp32").cuda()  # resnet50() initialises from the same seeded generator as the fixture
    m.train()
    data, target = synthetic_batch(2, S, seed=0, index=3)
    logits = m(data.cuda())
    loss = CrossEntropyLoss(smoothing=0.1)(logits, target.cuda())
    loss.backward()
    close(f"logits_{S}", logits, 1e-3, NET)  # north_star: fp32 logits within 1e-3 rel of the CPU forward
    assert abs(loss.item() - float(NET[f"loss_{S}"])) < 1e-4 * float(NET[f"loss_{S}"])
    close(f"bn1_running_var_{S}", dict(m.named_buffers())["bn1.running_var"], 1e-4, NET)
    close(f"fc_weight_grad_{S}", dict(m.named_parameters())["fc.weight"].grad[:8, :64], 1e-3, NET)


def test_loss_curve_golden(dev):
    """same seeds / schedule as make_golden.CURVE; the fp64-vs-fp32 gap of the stored oracle curves is the yardstick."""
    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD

    l32, l64, lrs = NET["curve_fp32"], NET["curve_fp64"], NET["curve_lrs"]
    m = resnet50(dtype="fp32").cuda()
    crit = CrossEntropyLoss(smoothing=0.1)
    opt = SGD([{"params": list(m.parameters())}], lr=0.0, momentum=0.9, weight_decay=3e-5)
    opt.attach_model(m)
    m.train()
    losses = []
    for i, lr in enumerate(lrs):
        data, target = synthetic_batch(8, 64, seed=0, index=i)
        for g in opt.param_groups:
            g["lr"] = float(lr)
        loss = crit(m(data.cuda()), target.cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    check_curve(losses, l32, l64, "fp32")


def _curve20(dtype):
    import numpy as np

    from sota_imagenet_amd.losses import CrossEntropyLoss
    from sota_imagenet_amd.models import resnet50
    from sota_imagenet_amd.optim import SGD

    with np.load(os.path.join(os.path.dirname(__file__), "golden", "curve20.npz")) as z:
        l32, l64, lrs = z["curve_fp32"], z["curve_fp64"], z["curve_lrs"]
    assert len(lrs) == 20 and abs(lrs[8] - 0.031) < 1e-12 and lrs.argmax() == 8 and abs(lrs[0] - 0.031e-3) < 1e-12  # 8 warm-up + 12 cosine steps
    m = resnet50(dtype=dtype).cuda()
    crit = CrossEntropyLoss(smoothing=0.1)
    opt = SGD([{"params": list(m.parameters())}], lr=0.0, momentum=0.9, weight_decay=3e-5)
    opt.attach_model(m)
    m.train()
    losses = []
    for i, lr in enumerate(lrs):
        data, target = synthetic_batch(32, 224, seed=0, index=i)
        for g in opt.param_groups:
            g["lr"] = float(lr)
        loss = crit(m(data.cuda()), target.cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    losses = np.asarray(losses, dtype=np.float64)
    dev_ = np.abs(losses - l64) / l64
    yard = np.abs(l32 - l64) / l64
    fmt = {"float_kind": lambda v: f"{v:.1e}"}
    print(f"curve20 [{dtype}] deviation from the fp64 oracle:", np.array2string(dev_, formatter=fmt, max_line_width=400), "yardstick (oracle fp32):",
          np.array2string(yard, formatter=fmt, max_line_width=400))
    return dev_, yard, losses, l64


def test_loss_curve_20_steps_golden(dev):
    """SURVEY 8(c)(iii) at the size it specifies: 20 SGD steps, bs 32, 224 px, seed 0, warm-up + cosine shape of the r50 recipe
    (1.r50_baseline.yaml:41-44) with the peak scaled to 0.031 — tests/golden/curve20.npz (make_golden.py curve20).  At this
    batch layer 4 normalises over 32 x 7 x 7 values and the curve is well conditioned: the oracle's own fp32 run stays within
    4e-3 of its fp64 run over all 20 steps.  Band: step 0 1e-5; steps 1-19 max(2 x that yardstick, 1e-2) of the fp64 curve."""
    import numpy as np

    dev_, yard, losses, l64 = _curve20("fp32")
    assert dev_[0] < 1e-5, (dev_[0], losses[0], l64[0])
    assert (dev_[1:] <= np.maximum(2 * yard[1:], 1e-2)).all(), (dev_, yard, losses, l64)


@pytest.mark.parametrize("dtype,band0,band", [("bf16", 5e-3, 3e-2), ("fp8", 5e-3, 3e-2)])
def test_loss_curve_20_steps_low_precision(dev, dtype, band0, band):
    """the same 20-step curve in the low-precision steps BASELINE's configs are quoted in: bf16 activations (configs[2]) and the
    fp8 convolution step (configs[4]; its first step is the bf16 calibration step).  Measured: bf16 within 1.0e-2 of the fp64 oracle
    curve over all 20 steps (mean 1e-3), fp8 within 6.6e-3 (mean 2.4e-3); band 3e-2 — a low-precision step that drifted from the recipe's
    curve (wrong scale, lost update, stale statistics) leaves it by far more."""
    dev_, _, losses, l64 = _curve20(dtype)
    assert dev_[0] < band0, (dev_[0], losses[0], l64[0])
    assert (dev_[1:] <= band).all(), (dev_, losses, l64)


def check_curve(losses, l32, l64, dtype):
    """The trajectory of a randomly initialised ResNet-50 on noise batches is chaotic: two correct fp32 implementations
    (torch CPU fp32 vs fp64 here) agree to ~1e-6 at step 0, ~1e-3 at step 1 and only to a few percent afterwards.  So:
    step 0 tight, step 1 moderately tight, later steps inside a band (6 % or 3x the reference's own fp32-vs-fp64 gap),
    and the mean deviation over the curve bounded."""
    import numpy as np

    losses, l32, l64 = (np.asarray(x, dtype=np.float64) for x in (losses, l32, l64))
    dev = np.abs(losses - l64) / l64
    yard = np.abs(l32 - l64) / l64
    t0, t1 = (1e-5, 3e-3) if dtype == "fp32" else (2e-2, 3e-2)
    assert dev[0] < t0, (dev[0], losses, l64)
    assert dev[1] < max(t1, 3 * yard[1]), (dev[1], losses, l64)
    assert (dev < np.maximum(3 * yard, 6e-2)).all(), (dev, yard, losses, l32, l64)
    assert dev.mean() < max(3 * yard.mean(), 3e-2), (dev.mean(), yard.mean())
