"""GPU parity of the generated assembly kernels (csrc/asm/dconv_gen.py, pw_gen.py) through the per-op C-ABI, at the batch the
BASELINE configs name (256 images): exact on small-integer data against torch's fp32 convolution of the same operands (any
summation order is exact there), forward with its BN-statistics rows and the data gradient; and on random data within one
bf16 rounding of the fp32 convolution.  The BN-backward-sums epilogue of the dgrad variants is exercised through the
executor (tests/test_resnet_gpu.py: test_teacher_forced_layers, test_baseline_batch_rule_selected_variants)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(256, 14, 256, 256, 3), (256, 7, 512, 512, 3), (256, 28, 128, 128, 3), (256, 56, 64, 64, 3), (256, 14, 256, 1024, 1), (256, 14, 1024, 256, 1),
          (256, 7, 2048, 512, 1), (256, 7, 512, 2048, 1), (256, 28, 512, 256, 1), (256, 14, 1024, 512, 1), (256, 28, 512, 128, 1)]


def _ref(x, w, K):
    return torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,Cin,Cout,K", SHAPES)
def test_forward_statistics_and_dgrad_are_exact_on_integer_data(dev, N, H, Cin, Cout, K):
    from sota_imagenet_amd import ops

    torch.manual_seed(0)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cout, K, K, Cin), device=dev).to(torch.bfloat16)
    y, part = ops.conv2d_fwd(x, w, 1, K // 2, stats=True)
    ref = _ref(x, w, K).to(torch.bfloat16)
    assert torch.equal(y, ref)
    assert part is not None and part.shape[1:] == (2, Cout)
    s1, s2 = ref.float().sum(dim=(0, 1, 2)).double(), (ref.float() ** 2).sum(dim=(0, 1, 2)).double()
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * s1.abs().max()
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * s2.abs().max()
    y0 = ops.conv2d_fwd(x, w, 1, K // 2)  # the variant without the statistics epilogue
    assert torch.equal(y0, ref)
    dy = torch.randint(-2, 3, (N, H, H, Cout), device=dev).to(torch.bfloat16)
    dx = ops.conv2d_dgrad(dy, w, (N, H, H, Cin), 1, K // 2)
    refd = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1)
    assert torch.equal(dx, refd.to(torch.bfloat16))


@pytest.mark.parametrize("N,H,Cin,Cout,K", SHAPES)
def test_forward_on_random_data_is_within_one_bf16_rounding(dev, N, H, Cin, Cout, K):
    from sota_imagenet_amd import ops

    torch.manual_seed(1)
    x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, K, K, Cin, device=dev) * 0.05).to(torch.bfloat16)
    y = ops.conv2d_fwd(x, w, 1, K // 2).float()
    ref = _ref(x, w, K)
    # tolerance: bf16 has 8 significant bits -> 2^-8 relative per element, plus fp32 summation-order noise on the largest values
    tol = 2.0 ** -8 * ref.abs() + 2.0 ** -9 * ref.abs().max() * 1e-2
    assert ((y - ref).abs() <= tol + 1e-6).all()


def test_other_batch_sizes_take_the_same_kernels(dev):
    """8 images: 8 tiles (layer 3) / 4 tiles (layer 4, two images each) / 14 pixel tiles of 112 (pointwise)"""
    from sota_imagenet_amd import ops

    torch.manual_seed(2)
    for (_, H, Cin, Cout, K) in SHAPES:
        x = torch.randint(-2, 3, (8, H, H, Cin), device=dev).to(torch.bfloat16)
        w = torch.randint(-2, 3, (Cout, K, K, Cin), device=dev).to(torch.bfloat16)
        y, part = ops.conv2d_fwd(x, w, 1, K // 2, stats=True)
        assert torch.equal(y, _ref(x, w, K).to(torch.bfloat16))


WG_SHAPES = [(256, 14, 256, 256), (256, 28, 128, 128), (256, 7, 512, 512), (256, 56, 64, 64), (8, 14, 256, 256), (12, 28, 128, 128), (20, 7, 512, 512),
             (5, 56, 64, 64), (64, 112, 64, 64)]


def _wgrad_ref(dy, x):
    """fp32 weight gradient of a 3x3 / pad 1 convolution, [Cout][3][3][Cin] (exact on small-integer data in any order)"""
    xp = torch.nn.functional.pad(x.float(), (0, 0, 1, 1, 1, 1))
    H, W = x.shape[1], x.shape[2]
    d2 = dy.float().reshape(-1, dy.shape[-1])
    out = torch.empty(dy.shape[-1], 3, 3, x.shape[-1], device=x.device)
    for ky in range(3):
        for kx in range(3):
            out[:, ky, kx] = d2.t() @ xp[:, ky:ky + H, kx:kx + W].reshape(-1, x.shape[-1])
    return out


@pytest.mark.parametrize("N,H,Cin,Cout", WG_SHAPES)
def test_weight_gradient_is_exact_on_integer_data(dev, N, H, Cin, Cout, monkeypatch):
    """the generated 3x3 weight-gradient kernels (csrc/asm/wg_gen.py) at the baseline batch and at batches that leave short last
    splits; the implicit-GEMM kernel (MI355_WG3=0 is read once per process, so it is compared through its own exactness test in
    test_ops_gpu.py) and this one must both equal the fp32 reference bit for bit on integer data"""
    from sota_imagenet_amd import ops

    torch.manual_seed(3)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    dy = torch.randint(-2, 3, (N, H, H, Cout), device=dev).to(torch.bfloat16)
    ref = _wgrad_ref(dy, x)
    dw = ops.conv2d_wgrad(dy, x, 3, 3, 1, 1)
    assert torch.equal(dw, ref)
    dw2 = ops.conv2d_wgrad(dy, x, 3, 3, 1, 1, dw=dw.clone(), beta=1.0)
    assert torch.equal(dw2, 2 * ref)


def test_weight_gradient_on_random_data(dev):
    from sota_imagenet_amd import ops

    torch.manual_seed(4)
    for (N, H, Cin, Cout) in WG_SHAPES[:4]:
        x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
        dy = (torch.randn(N, H, H, Cout, device=dev) * 0.1).to(torch.bfloat16)
        ref = _wgrad_ref(dy, x)
        dw = ops.conv2d_wgrad(dy, x, 3, 3, 1, 1)
        assert ((dw - ref).norm() / ref.norm()).item() < 1e-5  # fp32 accumulation of exact bf16 products: summation order only


@pytest.mark.parametrize("N,H,Cin,Cout", [(256, 14, 1024, 256), (256, 14, 256, 1024), (256, 7, 2048, 512), (256, 7, 512, 2048), (256, 28, 512, 128), (256, 28, 128, 512),
                                          (256, 56, 256, 64), (256, 56, 64, 256), (256, 56, 256, 128), (256, 28, 512, 256), (256, 14, 1024, 512),
                                          (3, 14, 256, 1024), (5, 7, 2048, 512), (3, 56, 64, 256), (3, 56, 256, 64), (3, 28, 512, 128)])
def test_pointwise_weight_gradient_is_exact_on_integer_data(dev, N, H, Cin, Cout):
    """the generated 1x1 weight-gradient kernels (csrc/asm/wg1_gen.py): every (channels, tile shape) variant at the baseline batch; flat
    64-pixel tiles with a ragged last tile (3 x 196 = 588 and 5 x 49 = 245 pixels are not multiples of 64); accumulation into an existing gradient"""
    from sota_imagenet_amd import ops

    torch.manual_seed(5)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    dy = torch.randint(-2, 3, (N, H, H, Cout), device=dev).to(torch.bfloat16)
    ref = (dy.float().reshape(-1, Cout).t() @ x.float().reshape(-1, Cin)).reshape(Cout, 1, 1, Cin)
    dw = ops.conv2d_wgrad(dy, x, 1, 1, 1, 0)
    assert torch.equal(dw, ref)
    assert torch.equal(ops.conv2d_wgrad(dy, x, 1, 1, 1, 0, dw=dw.clone(), beta=1.0), 2 * ref)


def test_reserved_cus_change_the_plans_not_the_results(dev):
    """mi355_set_reserved_cus: grids sized from the CU count plan for fewer CUs (other split counts for the generated weight-gradient
    kernels, fewer persistent workgroups); results stay exact.  mi355_comm_standin (the measurement stand-in of tools/reserve_cus_ab.py)
    launches and returns."""
    from sota_imagenet_amd import native, ops

    L = native.lib()
    torch.manual_seed(6)
    x = torch.randint(-2, 3, (64, 14, 14, 256), device=dev).to(torch.bfloat16)
    dy = torch.randint(-2, 3, (64, 14, 14, 256), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (1024, 1, 1, 256), device=dev).to(torch.bfloat16)
    ref = _wgrad_ref(dy, x)
    try:
        native.check(L.mi355_set_reserved_cus(32))
        assert torch.equal(ops.conv2d_wgrad(dy, x, 3, 3, 1, 1), ref)
        y = ops.conv2d_fwd(x, w, 1, 0)  # persistent pointwise kernel: units per workgroup from the CU count
        assert torch.equal(y, _ref(x, w, 1).to(torch.bfloat16))
        assert L.mi355_set_reserved_cus(12) != 0  # not a multiple of 8
    finally:
        native.check(L.mi355_set_reserved_cus(0))
    assert torch.equal(ops.conv2d_wgrad(dy, x, 3, 3, 1, 1), ref)
    native.check(L.mi355_comm_standin(4, 20, native.cur_stream()))
    torch.cuda.synchronize()
