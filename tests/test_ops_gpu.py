"""GPU parity of every per-op C-ABI entry point against the torch-CPU oracle (oracle/ops_ref.py).

Tolerances (normalised max error  max|got-ref| / max|ref|):
  fp32: 2e-5   (exact-fp32 MFMA fma chains; only the summation order differs from oneDNN)
  bf16: 2e-2   (inputs are rounded to bf16 before BOTH paths; accumulation is fp32 on both)
"""
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


def nerr(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def rnd(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(shape, generator=g) * 2 - 1) * scale
    return x.to(dtype).float()  # value representable in dtype, held as fp32 on the CPU side


CONV_CASES = [
    # N, H, W, Cin, Cout, K, stride
    (2, 14, 14, 64, 64, 1, 1),
    (2, 14, 14, 64, 128, 3, 1),
    (2, 14, 14, 128, 64, 3, 2),
    (2, 8, 8, 128, 256, 1, 2),
    (3, 7, 7, 256, 128, 3, 1),      # M=147: ragged last row tile
    (1, 56, 56, 64, 64, 3, 1),
    (2, 16, 16, 192, 320, 3, 2),    # Cout not a multiple of 128, Cin 3 k-slabs in fp32
    # mid-size: more items than persistent workgroups (several rounds per workgroup), 3-tap wgrad items with many splits
    (32, 56, 56, 64, 64, 3, 1),
    # 196 tiles of 256 rows, N % 256 == 0, K = 256: the 256x256 tile is chosen by the rule (bf16); wgrad: 4 tiles x 128
    # splits = 512 workgroups, XCD-local item order
    (64, 28, 28, 256, 256, 1, 1),
    (16, 28, 28, 128, 128, 3, 2),   # stride-2 3x3: four dgrad parity classes, several row tiles each
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, dtype, case):
    from sota_imagenet_amd import ops

    N, H, W, Cin, Cout, K, s = case
    pad = K // 2
    x = rnd((N, H, W, Cin), 1, dtype)
    w = rnd((Cout, K, K, Cin), 2, dtype, scale=0.1)
    y_ref = R.conv2d_fwd(x, w, s, pad)
    dy = rnd(tuple(y_ref.shape), 3, dtype)
    dx_ref, dw_ref = R.conv2d_bwd(x, w, dy, s, pad)
    add = rnd((N, H, W, Cin), 4, dtype)

    xd, wd, dyd, addd = (t.to(dev, dtype).contiguous() for t in (x, w, dy, add))
    y = ops.conv2d_fwd(xd, wd, s, pad)
    assert nerr(y, y_ref) < TOL[dtype], "fwd"
    dx = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, pad)
    assert nerr(dx, dx_ref) < TOL[dtype], "dgrad"
    dx2 = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, pad, addend=addd)
    assert nerr(dx2, dx_ref + add) < TOL[dtype], "dgrad+addend"
    dw = ops.conv2d_wgrad(dyd, xd, K, K, s, pad)
    assert nerr(dw, dw_ref) < TOL[dtype], "wgrad"
    # accumulate mode
    dw2 = ops.conv2d_wgrad(dyd, xd, K, K, s, pad, dw=dw.clone(), beta=1.0)
    assert nerr(dw2, 2 * dw_ref) < TOL[dtype], "wgrad beta=1"


def test_conv_exact_integers(dev):
    """A = asymmetric small integers: fp32 and bf16 results must be EXACT (catches any fragment-layout slip)."""
    from sota_imagenet_amd import ops

    N, H, W, Cin, Cout, K, s, pad = 2, 6, 6, 64, 128, 3, 1, 1
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-2, 3, (N, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, K, K, Cin), generator=g).float()
    y_ref = R.conv2d_fwd(x, w, s, pad)
    dy = torch.randint(-2, 3, tuple(y_ref.shape), generator=g).float()
    dx_ref, dw_ref = R.conv2d_bwd(x, w, dy, s, pad)
    for dtype in DTYPES:
        xd, wd, dyd = (t.to(dev, dtype).contiguous() for t in (x, w, dy))
        y = ops.conv2d_fwd(xd, wd, s, pad)
        assert torch.equal(y.float().cpu(), y_ref), f"fwd {dtype}"
        dx = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, pad)
        assert torch.equal(dx.float().cpu(), dx_ref.to(dtype).float()), f"dgrad {dtype}"
        dw = ops.conv2d_wgrad(dyd, xd, K, K, s, pad)
        assert torch.equal(dw.cpu(), dw_ref), f"wgrad {dtype}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 32), (1, 64, 96)])
def test_stem(dev, dtype, shape):
    from sota_imagenet_amd import ops

    N, H, W = shape
    x = rnd((N, 3, H, W), 11, dtype, scale=2.5)
    w = rnd((64, 7, 7, 3), 12, dtype, scale=0.2)
    x_nhwc = R.nchw_to_nhwc(x)
    y_ref = R.conv2d_fwd(x_nhwc, w, 2, 3)
    dy = rnd(tuple(y_ref.shape), 13, dtype)
    _, dw_ref = R.conv2d_bwd(x_nhwc, w, dy, 2, 3)
    xd = x.to(dev)
    xpad = ops.stem_ingest(xd, dtype)
    y = ops.stem_fwd(xpad, w.to(dev), N, H, W, dtype)
    assert nerr(y, y_ref) < TOL[dtype], "stem fwd"
    dw = ops.stem_wgrad(dy.to(dev, dtype), xpad, N, H, W)
    assert nerr(dw, dw_ref) < TOL[dtype], "stem wgrad"


@pytest.mark.parametrize("shape", [(2, 64, 64), (2, 96, 128), (1, 160, 160), (1, 224, 224), (1, 320, 320)])
def test_stem_direct_kernel_is_exact_on_integer_data(dev, shape, monkeypatch):
    """bf16 stem forward: the direct convolution out of raw input rows (csrc/stem_direct.hip stem_direct_kernel, the default) and the
    row-pair implicit GEMM (MI355_STEM_DIRECT=0) on small-integer data, where every fp32 partial sum is exact whatever the
    summation order: both must equal the oracle's result rounded to bf16 BIT for bit (tile shapes: 1 ... 10 fragments per row)."""
    from sota_imagenet_amd import ops

    N, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (N, 3, H, W), generator=g).float()
    w = torch.randint(-2, 3, (64, 7, 7, 3), generator=g).float()
    y_ref = R.conv2d_fwd(R.nchw_to_nhwc(x), w, 2, 3).to(torch.bfloat16).float()
    xpad = ops.stem_ingest(x.to(dev), torch.bfloat16)
    for direct in ("1", "0"):
        monkeypatch.setenv("MI355_STEM_DIRECT", direct)
        y = ops.stem_fwd(xpad, w.to(dev), N, H, W, torch.bfloat16)
        assert torch.equal(y.float().cpu(), y_ref), f"MI355_STEM_DIRECT={direct}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 7, 7, 64), (4, 14, 14, 256), (2, 5, 5, 2048), (3, 9, 11, 1024), (1, 28, 28, 128)])
@pytest.mark.parametrize("residual", [False, True])
def test_bn_train_fwd_bwd(dev, dtype, case, residual):
    from sota_imagenet_amd import ops

    N, H, W, C = case
    x = rnd((N, H, W, C), 21, dtype, scale=2.0) + 0.3
    x = x.to(dtype).float()
    gamma = rnd((C,), 22, torch.float32) + 1.5
    beta = rnd((C,), 23, torch.float32)
    rm = rnd((C,), 24, torch.float32)
    rv = rnd((C,), 25, torch.float32).abs() + 0.5
    res = rnd((N, H, W, C), 26, dtype) if residual else None
    dout = rnd((N, H, W, C), 27, dtype)
    out_ref, rm_ref, rv_ref, mean_ref, invstd_ref = R.bn_train(x, gamma, beta, rm, rv, res, True)
    dx_ref, dg_ref, db_ref, dres_ref = R.bn_train_bwd(x, gamma, beta, dout, res, True)

    d = lambda t, dt=torch.float32: None if t is None else t.to(dev, dt).contiguous()
    rmd, rvd = d(rm), d(rv)
    out, sm, si = ops.bn_fwd_train(d(x, dtype), d(gamma), d(beta), rmd, rvd, d(res, dtype), True)
    tol = TOL[dtype]
    assert nerr(out, out_ref) < max(tol, 1e-5), "out"
    assert nerr(sm, mean_ref) < 1e-5 and nerr(si, invstd_ref) < 1e-5, "saved stats"
    assert nerr(rmd, rm_ref) < 1e-5 and nerr(rvd, rv_ref) < 1e-5, "running stats"
    # backward uses the oracle's (fp32) activation for the mask so both sides mask identically
    out_for_mask = out_ref.to(dtype)
    dx, dg, db, dz = ops.bn_bwd(d(dout, dtype), d(out_for_mask, dtype), d(x, dtype), d(gamma), sm, si, True, want_dz=True)
    btol = 1e-4 if dtype == torch.float32 else 3e-2
    assert nerr(dx, dx_ref) < btol, "dx"
    assert nerr(dg, dg_ref) < btol and nerr(db, db_ref) < btol, "dgamma/dbeta"
    if residual:
        assert nerr(dz, dres_ref) < btol, "dz (residual gradient)"


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_eval(dev, dtype):
    from sota_imagenet_amd import ops

    N, H, W, C = 2, 7, 7, 512
    x = rnd((N, H, W, C), 31, dtype, scale=2.0)
    gamma, beta = rnd((C,), 32, torch.float32) + 1.5, rnd((C,), 33, torch.float32)
    rm, rv = rnd((C,), 34, torch.float32), rnd((C,), 35, torch.float32).abs() + 0.5
    ref = R.bn_eval(x, gamma, beta, rm, rv, None, True)
    out = ops.bn_fwd_eval(x.to(dev, dtype), gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), None, True)
    assert nerr(out, ref) < max(TOL[dtype], 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool(dev, dtype):
    from sota_imagenet_amd import ops

    N, H, W, C = 2, 16, 12, 64
    x = torch.relu(rnd((N, H, W, C), 41, dtype))  # post-ReLU input: many exact ties at 0
    y_ref, _ = R.maxpool(x)
    dy = rnd(tuple(y_ref.shape), 42, dtype)
    dx_ref = R.maxpool_bwd(x, dy)
    y, idx = ops.maxpool_fwd(x.to(dev, dtype))
    assert torch.equal(y.float().cpu(), y_ref)
    dx = ops.maxpool_bwd(dy.to(dev, dtype), idx, (N, H, W, C))
    assert nerr(dx, dx_ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gap(dev, dtype):
    from sota_imagenet_amd import ops

    N, H, W, C = 3, 7, 7, 2048
    x = rnd((N, H, W, C), 51, dtype)
    p = ops.gap_fwd(x.to(dev, dtype))
    assert nerr(p, R.gap(x)) < 1e-5
    for shape in [(2, 56, 56, 256), (5, 3, 5, 64), (1, 1, 1, 128)]:  # the maps ECA pools; HW below / not a multiple of the 32 pixel lanes
        xs = rnd(shape, 53, dtype)
        assert nerr(ops.gap_fwd(xs.to(dev, dtype)), R.gap(xs)) < 1e-5, shape
    dp = rnd((N, C), 52, torch.float32)
    dx = ops.gap_bwd(dp.to(dev), (N, H, W, C), dtype)
    assert nerr(dx, R.gap_bwd(dp, (N, H, W, C))) < TOL[dtype]


@pytest.mark.parametrize("smoothing", [0.0, 0.1])
@pytest.mark.parametrize("soft", [False, True])
def test_ce(dev, smoothing, soft):
    from sota_imagenet_amd import ops

    N, C = 37, 1000
    logits = rnd((N, C), 61, torch.float32, scale=6.0)
    g = torch.Generator().manual_seed(62)
    lab = torch.randint(0, C, (N,), generator=g)
    tgt = torch.nn.functional.one_hot(lab, C).float()
    if soft:  # mixup-style soft target
        lab2 = torch.randint(0, C, (N,), generator=g)
        tgt = 0.7 * tgt + 0.3 * torch.nn.functional.one_hot(lab2, C).float()
    loss_ref, dl_ref = R.smooth_ce_bwd(logits, tgt, smoothing)
    loss, dl = ops.ce_loss(logits.to(dev), tgt.to(dev), smoothing)
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
    assert nerr(dl, dl_ref) < 1e-5
    if smoothing and not soft:  # cross-check against torch's own label_smoothing on index targets
        l2 = torch.nn.functional.cross_entropy(logits, lab, label_smoothing=smoothing)
        assert abs(loss.item() - l2.item()) < 1e-5 * max(1.0, abs(l2.item()))


def test_sgd(dev):
    from sota_imagenet_amd import ops

    n = 100003  # not a multiple of 4: exercises the tail
    p0 = rnd((n,), 71, torch.float32)
    grads = [rnd((n,), 72 + i, torch.float32) for i in range(3)]
    p_ref, m_ref = R.sgd_steps(p0, grads, lr=0.1, momentum=0.9, weight_decay=3e-5)
    p = p0.to(dev).clone()
    m = torch.zeros_like(p)
    for g in grads:
        ops.sgd_step(p, g.to(dev), m, 0.1, 0.9, 3e-5)
    assert nerr(p, p_ref) < 1e-6 and nerr(m, m_ref) < 1e-6


def test_sgd_with_the_moving_average_in_the_same_pass(dev):
    """mi355_sgd_step_ema: the same parameters / momenta bit for bit as the plain step, and the average = what ModelEma's lerp after each
    step gives (ema + (1 - decay) * (p - ema), the callback's arithmetic: train.py:111-112)"""
    from sota_imagenet_amd import ops

    n = 100003
    p0 = rnd((n,), 71, torch.float32)
    grads = [rnd((n,), 72 + i, torch.float32).to(dev) for i in range(3)]
    p, pe = p0.to(dev).clone(), p0.to(dev).clone()
    m, me = torch.zeros_like(p), torch.zeros_like(p)
    ema, ema_ref = pe.clone(), p.clone()
    for g in grads:
        ops.sgd_step(p, g, m, 0.1, 0.9, 3e-5)
        ema_ref.lerp_(p, 1.0 - 0.99)
        ops.sgd_step(pe, g, me, 0.1, 0.9, 3e-5, ema=ema, ema_decay=0.99)
    assert torch.equal(p, pe) and torch.equal(m, me)
    assert nerr(ema, ema_ref) < 1e-6 and not torch.equal(ema, pe)
