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


# the other sizes of the progressive-resize recipe (BASELINE configs[4]: 160 / 320 px), at batches the plans of a 512-image step reduce to:
# (N, H, Cin, Cout, K, kernel family of the forward)
SHAPES_OTHER = [(32, 40, 64, 64, 3, "dconv_l1a"), (32, 20, 128, 128, 3, "dconv_l2a"), (64, 10, 256, 256, 3, "dconv_l3a"), (64, 5, 512, 512, 3, "dconv_l4a"),
                (8, 80, 64, 64, 3, "dconv_l1b"), (16, 40, 128, 128, 3, "dconv_l2b"), (32, 20, 256, 256, 3, "dconv_l3b"), (64, 10, 512, 512, 3, "dconv_l4b"),
                (64, 10, 1024, 256, 1, "pk_k1024_n256_w200"), (64, 5, 2048, 512, 1, "pk_k2048_n512_w100"), (32, 20, 512, 128, 1, "po_k512_b128"),
                (64, 10, 1024, 512, 1, "pk_k1024_n512_w200"),
                # BASELINE configs[3] (anti-aliased BResNet-50): conv2 of the striding blocks at their input resolution, the deep stem's 3x3s at 112 x 112
                (16, 112, 64, 64, 3, "dconv_v0"), (32, 56, 128, 128, 3, "dconv_v2"), (64, 28, 256, 256, 3, "dconv_v3"), (256, 14, 512, 512, 3, "dconv_v4")]


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


@pytest.mark.parametrize("N,H,Cin,Cout,K,fam", SHAPES_OTHER)
def test_the_160_and_320_px_shapes_take_generated_kernels_and_are_exact(dev, N, H, Cin, Cout, K, fam):
    from sota_imagenet_amd import ops

    torch.manual_seed(12)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cout, K, K, Cin), device=dev).to(torch.bfloat16)
    y, part = ops.conv2d_fwd(x, w, 1, K // 2, stats=True)
    assert ops.last_conv_kernel().startswith(fam + "_s1"), ops.last_conv_kernel()
    ref = _ref(x, w, K).to(torch.bfloat16)
    assert torch.equal(y, ref)
    s1, s2 = ref.float().sum(dim=(0, 1, 2)).double(), (ref.float() ** 2).sum(dim=(0, 1, 2)).double()
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * s1.abs().max()
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * s2.abs().max()
    dy = torch.randint(-2, 3, (N, H, H, Cout), device=dev).to(torch.bfloat16)
    dx = ops.conv2d_dgrad(dy, w, (N, H, H, Cin), 1, K // 2)
    refd = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1)
    assert torch.equal(dx, refd.to(torch.bfloat16))
    if K == 3:
        assert ops.last_conv_kernel().startswith(fam + "_s0"), ops.last_conv_kernel()


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


def _pack_bits(mask):
    """bool [..., C] -> uint8 [..., C / 8]: bit e of byte j = channel 8j + e (the ReLU bit masks bn_apply writes)"""
    m = mask.reshape(*mask.shape[:-1], -1, 8).to(torch.int32)
    return (m << torch.arange(8, device=mask.device, dtype=torch.int32)).sum(-1).to(torch.uint8)


# (batch, H, channels of dx = 4 * channels of dy ... , channels of dy, kernel family): conv1's data gradient of every bottleneck shape of the
# network at the BASELINE batch, the first blocks (dy at the previous stage's resolution), and batches whose pixel count is not a multiple
# of the 64-pixel tile
PO_DGRAD = [(256, 56, 256, 64, "po_k64_b256"), (256, 56, 256, 128, "po_k128_b256"), (256, 28, 512, 128, "po_k128_b256"), (256, 28, 512, 256, "po_k256_b256"),
            (256, 14, 1024, 256, "po_k256_b256"), (256, 14, 1024, 512, "po_k512_b128"), (256, 7, 2048, 512, "po_k512_b128"),
            (3, 14, 1024, 256, "po_k256_b256"), (5, 7, 2048, 512, "po_k512_b128"), (1, 56, 256, 64, "po_k64_b256"),
            # into 64 columns (waves 2 x 2, 128-pixel tiles): conv3's / the downsample conv's data gradient of layer 1, layer1.0's conv1
            (256, 56, 64, 256, "po_k256_b64"), (256, 56, 64, 64, "po_k64_b64"), (3, 20, 64, 256, "po_k256_b64"), (1, 10, 64, 64, "po_k64_b64")]


@pytest.mark.parametrize("N,H,Cx,Cy,fam", PO_DGRAD)
@pytest.mark.parametrize("add,sums", [(2, True), (1, True), (0, True), (2, False), (1, False)])
def test_conv1_data_gradient_with_shortcut_addend_and_bn_backward_sums_is_exact(dev, N, H, Cx, Cy, fam, add, sums, monkeypatch):
    monkeypatch.setenv("MI355_PO", "2")  # every variant, also where the default rule keeps a launch on another kernel
    """the generated output-heavy pointwise kernels (csrc/asm/po_gen.py) through mi355_conv2d_dgrad_bn, the launch form of the executor's
    backward: dx = dgrad(dy) + shortcut gradient (plain / under the block output's ReLU bits), bit for bit against the fp32 reference on
    integer data, and the BN-backward partial rows (sum dz, sum dz * xhat) against fp64 sums over the same tensors; the kernel that ran
    is asserted by name (a selection regression must fail here, not in a profile)"""
    from sota_imagenet_amd import ops

    torch.manual_seed(7)
    dy = torch.randint(-2, 3, (N, H, H, Cy), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cy, 1, 1, Cx), device=dev).to(torch.bfloat16)   # forward weights [Cout = Cy][1][1][Cin = Cx]
    ad = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16) if add else None
    abits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8) if add == 2 else None
    y = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16) if sums else None
    ybits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8) if sums else None
    mean = (torch.randint(-4, 5, (Cx,), device=dev) * 0.25).float() if sums else None
    invstd = (torch.randint(1, 5, (Cx,), device=dev) * 0.5).float() if sums else None
    dx, part = ops.conv2d_dgrad_bn(dy, w, (N, H, H, Cx), 1, 0, addend=ad, addend_bits=abits, bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd)
    assert ops.last_conv_kernel() == "%s_s%d_a%d" % (fam, 2 if sums else 0, add)
    _check_dgrad_bn(dev, dx, part, dy, w, ad, abits, y, ybits, mean, invstd, add, sums)


def _check_dgrad_bn(dev, dx, part, dy, w, ad, abits, y, ybits, mean, invstd, add, sums):
    Cy, Cx = w.shape[0], w.shape[3]
    ref = dy.float().reshape(-1, Cy) @ w.float().reshape(Cy, Cx)
    if add:
        a = ad.float().reshape(-1, Cx)
        if add == 2:
            bit = ((abits.reshape(-1, Cx // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(-1, Cx)
            a = a * bit
        ref = ref + a
    ref = ref.to(torch.bfloat16)
    assert torch.equal(dx.reshape(-1, Cx), ref)
    if sums:
        assert part is not None and part.shape[1:] == (2, Cx)
        bit = ((ybits.reshape(-1, Cx // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(-1, Cx)
        dz = ref.double() * bit
        xhat = (y.double().reshape(-1, Cx) - mean.double()) * invstd.double()
        s1, s2 = dz.sum(0), (dz * xhat).sum(0)
        assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * max(1.0, s1.abs().max().item())
        assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * max(1.0, s2.abs().max().item())
    else:
        assert part is None


@pytest.mark.parametrize("N,H,Cx,Cy,fam", [(256, 56, 256, 128, "po_k128_b256"), (256, 28, 512, 256, "po_k256_b256"), (256, 14, 1024, 512, "po_k512_b128"), (6, 28, 512, 256, "po_k256_b256")])
@pytest.mark.parametrize("sums", [True, False])
def test_conv1_data_gradient_with_the_downsample_gradient_at_half_resolution(dev, N, H, Cx, Cy, fam, sums):
    """the first block of a stage: conv1's data gradient adds the downsample branch's gradient, which exists only at the even pixels
    (stride-2 1x1 convolution): the kernel takes it as the compact [N][H/2][W/2][C] tensor (addend_sub2) — same bits as adding the
    zero-filled full-size tensor"""
    from sota_imagenet_amd import ops

    torch.manual_seed(13)
    dy = torch.randint(-2, 3, (N, H, H, Cy), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cy, 1, 1, Cx), device=dev).to(torch.bfloat16)
    adc = torch.randint(-3, 4, (N, H // 2, H // 2, Cx), device=dev).to(torch.bfloat16)
    full = torch.zeros((N, H, H, Cx), device=dev, dtype=torch.bfloat16)
    full[:, ::2, ::2] = adc
    y = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16) if sums else None
    ybits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8) if sums else None
    mean = (torch.randint(-4, 5, (Cx,), device=dev) * 0.25).float() if sums else None
    invstd = (torch.randint(1, 5, (Cx,), device=dev) * 0.5).float() if sums else None
    dx, part = ops.conv2d_dgrad_bn(dy, w, (N, H, H, Cx), 1, 0, addend=adc, addend_sub2=True, bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd)
    assert ops.last_conv_kernel() == "%s_s%d_a3" % (fam, 2 if sums else 0)
    _check_dgrad_bn(dev, dx, part, dy, w, full, None, y, ybits, mean, invstd, 1, sums)


# (N, H, channels of dx, channels of dy, kernel size, family): conv2's data gradient (3x3: the BN-backward sums of bn1) and conv3's (long 1x1 reduction: bn2)
S2_OTHER = [(256, 56, 64, 64, 3, "dconv_l1"), (256, 28, 128, 128, 3, "dconv_l2"), (256, 14, 256, 256, 3, "dconv_l3"), (256, 7, 512, 512, 3, "dconv_l4"),
            (256, 14, 256, 1024, 1, "pk_k1024_n256_w196"), (256, 7, 512, 2048, 1, "pk_k2048_n512_w98"), (256, 28, 128, 512, 1, "po_k512_b128"),
            (64, 10, 256, 256, 3, "dconv_l3a"), (64, 5, 512, 512, 3, "dconv_l4a"), (16, 40, 128, 128, 3, "dconv_l2b"), (8, 80, 64, 64, 3, "dconv_l1b"),
            (64, 10, 256, 1024, 1, "pk_k1024_n256_w200")]


@pytest.mark.parametrize("N,H,Cx,Cy,K,fam", S2_OTHER)
def test_bn_backward_sums_epilogue_of_the_3x3_and_long_reduction_kernels_is_exact(dev, N, H, Cx, Cy, K, fam):
    """every `_s2` variant of dconv / pk (and po without an addend) through mi355_conv2d_dgrad_bn: dx bit for bit, the partial rows against
    fp64 sums of the same integer tensors (round 4 checked these epilogues only through the executor)"""
    from sota_imagenet_amd import ops

    torch.manual_seed(14)
    dy = torch.randint(-2, 3, (N, H, H, Cy), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cy, K, K, Cx), device=dev).to(torch.bfloat16)
    y = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16)
    ybits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8)
    mean, invstd = (torch.randint(-4, 5, (Cx,), device=dev) * 0.25).float(), (torch.randint(1, 5, (Cx,), device=dev) * 0.5).float()
    dx, part = ops.conv2d_dgrad_bn(dy, w, (N, H, H, Cx), 1, K // 2, bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd)
    assert ops.last_conv_kernel().startswith(fam + "_s2"), ops.last_conv_kernel()
    ref = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(dx, ref)
    bit = ((ybits.reshape(-1, Cx // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(-1, Cx)
    dz = ref.reshape(-1, Cx).double() * bit
    xhat = (y.double().reshape(-1, Cx) - mean.double()) * invstd.double()
    s1, s2 = dz.sum(0), (dz * xhat).sum(0)
    assert part is not None and part.shape[1:] == (2, Cx)
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * max(1.0, s1.abs().max().item())
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * max(1.0, s2.abs().max().item())


# (N, H of dx, channels, family): conv2 of layer2.0 / layer3.0 / layer4.0 — the stride sits in the 3x3 (torchvision v1.5) — and a ragged batch
S2D = [(256, 56, 128, "dconv_l2_d2"), (256, 28, 256, "dconv_l3_d2"), (256, 14, 512, "dconv_l4_d2"), (6, 28, 256, "dconv_l3_d2"), (2, 14, 512, "dconv_l4_d2")]


@pytest.mark.parametrize("N,H,C,fam", S2D)
@pytest.mark.parametrize("sums", [True, False])
def test_stride2_3x3_data_gradient_by_output_parity_classes_is_exact(dev, N, H, C, fam, sums):
    """the generated data gradient of the stride-2 3x3 convolutions (asm/dconv_gen.py Cfg.s2d: one workgroup per (tile, output parity class), the staged
    dy tile read at the class's taps) through mi355_conv2d_dgrad_bn: dx bit for bit against torch's transposed convolution on integer data, the
    BN-backward partial rows (one per tile and class) against fp64 sums; MI355_DCONV_S2=0 is the implicit-GEMM kernel on the same operands"""
    from sota_imagenet_amd import ops

    torch.manual_seed(21)
    Ho = H // 2
    dy = torch.randint(-2, 3, (N, Ho, Ho, C), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (C, 3, 3, C), device=dev).to(torch.bfloat16)
    y = torch.randint(-3, 4, (N, H, H, C), device=dev).to(torch.bfloat16)
    ybits = torch.randint(0, 256, (N, H, H, C // 8), device=dev, dtype=torch.uint8)
    mean, invstd = (torch.randint(-4, 5, (C,), device=dev) * 0.25).float(), (torch.randint(1, 5, (C,), device=dev) * 0.5).float()
    kw = dict(bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd) if sums else {}
    dx, part = ops.conv2d_dgrad_bn(dy, w, (N, H, H, C), 2, 1, **kw)
    assert ops.last_conv_kernel() == fam + ("_s2" if sums else "_s0"), ops.last_conv_kernel()
    ref = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=2, padding=1, output_padding=1)
    ref = ref.permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(dx, ref)
    if sums:
        bit = ((ybits.reshape(-1, C // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(-1, C)
        dz = ref.reshape(-1, C).double() * bit
        xhat = (y.double().reshape(-1, C) - mean.double()) * invstd.double()
        s1, s2 = dz.sum(0), (dz * xhat).sum(0)
        assert part is not None and part.shape[1:] == (2, C)
        assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * max(1.0, s1.abs().max().item())
        assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * max(1.0, s2.abs().max().item())


def test_stride2_data_gradient_switch_keeps_the_implicit_gemm_kernel(dev, monkeypatch):
    from sota_imagenet_amd import native, ops

    torch.manual_seed(22)
    N, H, C = 4, 28, 256
    dy = torch.randint(-2, 3, (N, H // 2, H // 2, C), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (C, 3, 3, C), device=dev).to(torch.bfloat16)
    a, _ = ops.conv2d_dgrad_bn(dy, w, (N, H, H, C), 2, 1)
    assert ops.last_conv_kernel() == "dconv_l3_d2_s0"
    monkeypatch.setenv("MI355_DCONV_S2", "0")
    native.lib().mi355_reload_knobs()
    b, _ = ops.conv2d_dgrad_bn(dy, w, (N, H, H, C), 2, 1)
    assert not ops.last_conv_kernel().startswith("dconv"), ops.last_conv_kernel()
    assert torch.equal(a, b)


# (N, H, C, family): conv2 of the stride-1 bottlenecks of layers 1 .. 4 (C -> C), + ragged batches
BN_IN = [(256, 56, 64, "dconv_l1"), (256, 28, 128, "dconv_l2"), (256, 14, 256, "dconv_l3"), (256, 7, 512, "dconv_l4"), (3, 28, 128, "dconv_l2"), (6, 7, 512, "dconv_l4")]


@pytest.mark.parametrize("N,H,C,fam", BN_IN)
def test_conv2_forward_with_bn1_in_its_operand_path_is_exact(dev, N, H, C, fam, monkeypatch):
    """dconv_*_s1_bn (asm/dconv_gen.py Cfg.bnin; MI355_DCONV_BN=1: off by default, measured break-even) through mi355_conv2d_fwd_bn_in on dyadic data,
    where every step is exact: the activation a = relu(y * scale + shift) and its ReLU bits the launch leaves behind, y2 = conv(a, w) bit for bit against
    torch's fp32 convolution, and the BatchNorm statistics rows of y2"""
    from sota_imagenet_amd import native, ops

    monkeypatch.setenv("MI355_DCONV_BN", "1")
    native.lib().mi355_reload_knobs()
    torch.manual_seed(23)
    y1 = torch.randint(-3, 4, (N, H, H, C), device=dev).to(torch.bfloat16)
    scale = (torch.randint(1, 9, (C,), device=dev) * 0.25).float()
    shift = (torch.randint(-8, 9, (C,), device=dev) * 0.25).float()
    w = torch.randint(-2, 3, (C, 3, 3, C), device=dev).to(torch.bfloat16)
    y2, part, a, bits = ops.conv2d_fwd_bn_in(y1, scale, shift, w)
    assert ops.last_conv_kernel() == fam + "_s1_bn", ops.last_conv_kernel()
    v = y1.float() * scale + shift
    a_ref = v.clamp_min(0).to(torch.bfloat16)
    assert torch.equal(a, a_ref)
    bits_ref = ((v > 0).reshape(N, H, H, C // 8, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, bits_ref)
    ref = _ref(a_ref, w, 3).to(torch.bfloat16)
    assert torch.equal(y2, ref)
    s1, s2 = ref.double().sum(dim=(0, 1, 2)), (ref.double() ** 2).sum(dim=(0, 1, 2))
    assert part is not None and part.shape[1:] == (2, C)
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * max(1.0, s1.abs().max().item())
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * s2.abs().max().item()


def test_conv2_forward_with_bn1_in_its_operand_path_against_the_oracle_and_the_unfused_launches(dev, monkeypatch):
    """random data: (i) against oracle/ops_ref: BatchNorm (batch statistics) -> ReLU -> conv, with the coefficients mi355_bn_fwd_train leaves;
    (ii) bit for bit against the library's own two launches (bn_apply, then the plain forward kernel)"""
    from oracle import ops_ref
    from sota_imagenet_amd import native, ops

    monkeypatch.setenv("MI355_DCONV_BN", "1")
    native.lib().mi355_reload_knobs()
    torch.manual_seed(24)
    N, H, C = 8, 14, 256
    y1 = (torch.randn(N, H, H, C) * 2).to(torch.bfloat16)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.3
    w = (torch.randn(C, 3, 3, C) * 0.05).to(torch.bfloat16)
    a_o, _, _, mean, invstd = ops_ref.bn_train(y1, gamma, beta, torch.zeros(C), torch.ones(C), relu=True)
    out_o = ops_ref.conv2d_fwd(a_o.to(torch.bfloat16), w, 1, 1)
    scale = (gamma * invstd).float()
    shift = (beta - mean * scale).float()
    y2, part, a, bits = ops.conv2d_fwd_bn_in(y1.to(dev), scale.to(dev), shift.to(dev), w.to(dev))
    assert ops.last_conv_kernel() == "dconv_l3_s1_bn"
    assert (a.float().cpu() - a_o.float()).abs().max() <= 2.0 ** -7 * a_o.abs().max()        # one bf16 rounding of the larger values
    err = (y2.float().cpu() - out_o.float()).abs().max() / out_o.float().abs().max()
    assert err < 2e-2, err
    a_n = torch.relu(torch.addcmul(shift.to(dev), y1.to(dev).float(), scale.to(dev))).to(torch.bfloat16)   # fma, one rounding: bn_apply's arithmetic
    y_plain, _ = ops.conv2d_fwd(a, w.to(dev), 1, 1, stats=True)
    assert ops.last_conv_kernel() == "dconv_l3_s1"
    assert torch.equal(y2, y_plain)
    assert (a.float() - a_n.float()).abs().max() <= 2.0 ** -7 * a_n.float().abs().max()


# (N, H, Cin, Cout, family): conv3 of the bottlenecks of layers 1 .. 3 (bn2 + ReLU in its operand path), + a small batch of whole 64-pixel tiles
PO_BN = [(256, 56, 64, 256, "po_k64_b256"), (256, 28, 128, 512, "po_k128_b256"), (256, 14, 256, 1024, "po_k256_b256"), (4, 28, 128, 512, "po_k128_b256")]


@pytest.mark.parametrize("N,H,Cin,Cout,fam", PO_BN)
def test_conv3_forward_with_bn2_in_its_operand_path_is_exact(dev, N, H, Cin, Cout, fam):
    """po_*_s1_a0_bn (asm/po_gen.py PoCfg.bnin) through mi355_conv2d_fwd_bn_in on dyadic data: the activation and its ReLU bits the launch leaves behind,
    y3 = conv1x1(a, w) bit for bit, the BatchNorm statistics rows of y3"""
    from sota_imagenet_amd import ops

    torch.manual_seed(25)
    y2 = torch.randint(-3, 4, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    scale = (torch.randint(1, 9, (Cin,), device=dev) * 0.25).float()
    shift = (torch.randint(-8, 9, (Cin,), device=dev) * 0.25).float()
    w = torch.randint(-2, 3, (Cout, 1, 1, Cin), device=dev).to(torch.bfloat16)
    y3, part, a, bits = ops.conv2d_fwd_bn_in(y2, scale, shift, w, pad=0)
    assert ops.last_conv_kernel() == fam + "_s1_a0_bn", ops.last_conv_kernel()
    v = y2.float() * scale + shift
    a_ref = v.clamp_min(0).to(torch.bfloat16)
    assert torch.equal(a, a_ref)
    bits_ref = ((v > 0).reshape(N, H, H, Cin // 8, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, bits_ref)
    ref = (a_ref.float().reshape(-1, Cin) @ w.float().reshape(Cout, Cin).t()).reshape(N, H, H, Cout).to(torch.bfloat16)
    assert torch.equal(y3, ref)
    s1, s2 = ref.double().sum(dim=(0, 1, 2)), (ref.double() ** 2).sum(dim=(0, 1, 2))
    assert part is not None and part.shape[1:] == (2, Cout)
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * max(1.0, s1.abs().max().item())
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * s2.abs().max().item()


def test_conv3_forward_with_bn2_in_its_operand_path_random_data_and_ragged_pixel_counts(dev):
    """random data against the library's own two launches (bn_apply's arithmetic, then the plain kernel); a pixel count that is no multiple of the 64-pixel
    tile has no kernel of this form: the entry point says so (the executor then runs bn_apply)"""
    from sota_imagenet_amd import native, ops

    torch.manual_seed(26)
    N, H, Cin, Cout = 8, 28, 128, 512
    y2 = (torch.randn(N, H, H, Cin, device=dev) * 2).to(torch.bfloat16)
    scale, shift = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    w = (torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05).to(torch.bfloat16)
    y3, part, a, bits = ops.conv2d_fwd_bn_in(y2, scale, shift, w, pad=0)
    assert ops.last_conv_kernel() == "po_k128_b256_s1_a0_bn"
    a_n = torch.relu(torch.addcmul(shift, y2.float(), scale)).to(torch.bfloat16)
    assert (a.float() - a_n.float()).abs().max() <= 2.0 ** -7 * a_n.float().abs().max()
    y_plain, _ = ops.conv2d_fwd(a, w, 1, 0, stats=True)
    assert ops.last_conv_kernel() == "po_k128_b256_s1_a0"
    assert torch.equal(y3, y_plain)
    assert torch.equal(bits, ((a.float() != 0).reshape(N, H, H, Cin // 8, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8))
    with pytest.raises(RuntimeError, match="no kernel applies the input's BatchNorm"):
        ops.conv2d_fwd_bn_in(y2[:3], scale, shift, w, pad=0)      # 3 * 784 pixels: not whole 64-pixel tiles


def test_bn_backward_sums_of_the_data_gradient_against_the_oracle(dev):
    """the same launch against oracle/ops_ref at a batch the CPU oracle handles: y -> BatchNorm (batch statistics) -> ReLU, the data
    gradient of the next conv as the gradient of that activation; the partial rows must add up to the oracle's dbeta / dgamma / (gamma = 1)"""
    from oracle import ops_ref
    from sota_imagenet_amd import ops

    torch.manual_seed(8)
    N, H, Cx, Cy = 8, 14, 1024, 256
    y = torch.randint(-3, 4, (N, H, H, Cx)).to(torch.bfloat16)
    gamma, beta = torch.ones(Cx), (torch.randint(-2, 3, (Cx,)) * 0.25).float()
    out, _, _, mean, invstd = ops_ref.bn_train(y, gamma, beta, torch.zeros(Cx), torch.ones(Cx), relu=True)
    dy = torch.randint(-2, 3, (N, H, H, Cy)).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cy, 1, 1, Cx)).to(torch.bfloat16)
    g_ref = (dy.float().reshape(-1, Cy) @ w.float().reshape(Cy, Cx)).reshape(N, H, H, Cx)  # exact integers: the activation gradient
    _, dgamma, dbeta, _ = ops_ref.bn_train_bwd(y, gamma, beta, g_ref, relu=True)
    bits = _pack_bits(out > 0)
    dx, part = ops.conv2d_dgrad_bn(dy.to(dev), w.to(dev), (N, H, H, Cx), 1, 0, bn_y=y.to(dev), bn_bits=bits.to(dev), bn_mean=mean.to(dev), bn_invstd=invstd.to(dev))
    assert ops.last_conv_kernel() == "po_k256_b256_s2_a0"
    assert torch.equal(dx.cpu().float(), g_ref)
    s = part.double().sum(0).cpu()
    assert (s[0] - dbeta.double()).abs().max() <= 1e-4 * dbeta.abs().max()
    assert (s[1] - dgamma.double()).abs().max() <= 1e-4 * dgamma.abs().max()


# BASELINE configs[3] (BResNet-50): every data gradient whose output is the gradient of a leaky-ReLU activation behind a BatchNorm —
# (N, H, Cx = channels of dx, Cy = channels of dy, K, kernel): conv2's 3x3 (stride-1 blocks, striding blocks at their input resolution, the deep stem),
# conv3's 1x1 (4 C -> C)
LEAKY = [(256, 56, 64, 64, 3, "dconv_l1_s3"), (256, 28, 128, 128, 3, "dconv_l2_s3"), (256, 14, 256, 256, 3, "dconv_l3_s3"), (256, 7, 512, 512, 3, "dconv_l4_s3"),
         (32, 56, 128, 128, 3, "dconv_v2_s3"), (64, 28, 256, 256, 3, "dconv_v3_s3"), (256, 14, 512, 512, 3, "dconv_v4_s3"), (16, 112, 64, 64, 3, "dconv_v0_s3"),
         (256, 56, 64, 256, 1, "po_k256_b64_s3_a0"), (256, 28, 128, 512, 1, "po_k512_b128_s3_a0"), (256, 14, 256, 1024, 1, "pk_k1024_n256_w196_s3"),
         (256, 7, 512, 2048, 1, "pk_k2048_n512_w98_s3"), (6, 14, 256, 1024, 1, "pk_k1024_n256_w196_s3")]


@pytest.mark.parametrize("N,H,Cx,Cy,K,kernel", LEAKY)
def test_data_gradient_with_bn_backward_sums_under_a_leaky_mask_is_exact(dev, N, H, Cx, Cy, K, kernel):
    """mi355_conv2d_dgrad_bn_leaky (slope 0.01): dx bit for bit the fp32 transposed convolution of integer data, the partial rows against fp64 sums of
    dz = dx where the bit is set, fp32(dx) * fp32(0.01) elsewhere (what bn_reduce_kernel<MASK = 3> forms), and dz * xhat; the kernel asserted by name"""
    from sota_imagenet_amd import ops

    torch.manual_seed(21)
    dy = torch.randint(-2, 3, (N, H, H, Cy), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cy, K, K, Cx), device=dev).to(torch.bfloat16)
    y = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16)
    ybits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8)
    mean = (torch.randint(-4, 5, (Cx,), device=dev) * 0.25).float()
    invstd = (torch.randint(1, 5, (Cx,), device=dev) * 0.5).float()
    dx, part = ops.conv2d_dgrad_bn(dy, w, (N, H, H, Cx), 1, K // 2, bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd, slope=0.01)
    assert ops.last_conv_kernel() == kernel
    ref = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=K // 2).permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(dx, ref)
    bit = ((ybits.reshape(-1, Cx // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(-1, Cx)
    r32 = ref.float().reshape(-1, Cx)
    dz = torch.where(bit > 0, r32, r32 * 0.01).double()
    xhat = (y.double().reshape(-1, Cx) - mean.double()) * invstd.double()
    s1, s2 = dz.sum(0), (dz * xhat).sum(0)
    assert part is not None and part.shape[1:] == (2, Cx)
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 2e-6 * max(1.0, s1.abs().max().item())
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 2e-6 * max(1.0, s2.abs().max().item())


def test_a_leaky_slope_without_a_generated_kernel_is_an_error(dev):
    """no other kernel has the leaky epilogue: a slope other than 0.01, fp32, or a shape without a generated kernel fail the call (the caller runs
    the BatchNorm backward's own reduction pass) instead of silently returning ReLU sums"""
    from sota_imagenet_amd import ops

    N, H, Cx, Cy = 8, 14, 256, 1024
    dy = torch.zeros((N, H, H, Cy), device=dev, dtype=torch.bfloat16)
    w = torch.zeros((Cy, 1, 1, Cx), device=dev, dtype=torch.bfloat16)
    y = torch.zeros((N, H, H, Cx), device=dev, dtype=torch.bfloat16)
    ybits = torch.zeros((N, H, H, Cx // 8), device=dev, dtype=torch.uint8)
    mean, invstd = torch.zeros(Cx, device=dev), torch.ones(Cx, device=dev)
    with pytest.raises(RuntimeError, match="leaky"):
        ops.conv2d_dgrad_bn(dy, w, (N, H, H, Cx), 1, 0, bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd, slope=0.2)
    with pytest.raises(RuntimeError, match="leaky"):
        ops.conv2d_dgrad_bn(dy.float(), w.float(), (N, H, H, Cx), 1, 0, bn_y=y.float(), bn_bits=ybits, bn_mean=mean, bn_invstd=invstd, slope=0.01)
    with pytest.raises(RuntimeError, match="leaky"):   # 18 x 18 pixels: no generated kernel
        ops.conv2d_dgrad_bn(dy[:, :9, :9].repeat(1, 2, 2, 1).contiguous(), w, (N, 18, 18, Cx), 1, 0, bn_y=y[:, :9, :9].repeat(1, 2, 2, 1).contiguous(),
                            bn_bits=ybits[:, :9, :9].repeat(1, 2, 2, 1).contiguous(), bn_mean=mean, bn_invstd=invstd, slope=0.01)


@pytest.mark.parametrize("N,H,Cin,Cout,fam", [(256, 56, 64, 256, "po_k64_b256"), (256, 28, 128, 512, "po_k128_b256"), (7, 28, 128, 512, "po_k128_b256"),
                                                (256, 56, 256, 128, "po_k256_b128"), (5, 20, 256, 128, "po_k256_b128"),
                                                (256, 56, 256, 64, "po_k256_b64"), (256, 56, 64, 64, "po_k64_b64"), (3, 10, 256, 64, "po_k256_b64")])
def test_conv3_forward_of_layers_1_and_2_takes_the_resident_weight_kernel(dev, N, H, Cin, Cout, fam):
    from sota_imagenet_amd import ops

    torch.manual_seed(9)
    x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cout, 1, 1, Cin), device=dev).to(torch.bfloat16)
    y, part = ops.conv2d_fwd(x, w, 1, 0, stats=True)
    assert ops.last_conv_kernel() == fam + "_s1_a0"
    ref = _ref(x, w, 1).to(torch.bfloat16)
    assert torch.equal(y, ref)
    s1, s2 = ref.float().sum(dim=(0, 1, 2)).double(), (ref.float() ** 2).sum(dim=(0, 1, 2)).double()
    assert (part[:, 0].double().sum(0) - s1).abs().max() <= 1e-6 * s1.abs().max()
    assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * s2.abs().max()
    assert torch.equal(ops.conv2d_fwd(x, w, 1, 0), ref) and ops.last_conv_kernel() == fam + "_s0_a0"


def test_the_default_rule_keeps_two_launch_shapes_on_the_older_kernels(dev):
    """MI355_PO=1 (default): layer 1's conv1 data gradient (shortcut addend + BN-backward sums) stays on the implicit-GEMM kernel, layer 4's
    conv3 forward on the long-reduction kernel — and their results are the same bits"""
    from sota_imagenet_amd import ops

    torch.manual_seed(11)
    for (N, H, Cx, Cy, want) in ((64, 56, 256, 64, "igemm<bf16,"), (64, 7, 2048, 512, "po_k512_b128_s2_a2")):
        dy = torch.randint(-2, 3, (N, H, H, Cy), device=dev).to(torch.bfloat16)
        w = torch.randint(-2, 3, (Cy, 1, 1, Cx), device=dev).to(torch.bfloat16)
        ad = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16)
        abits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8)
        y = torch.randint(-3, 4, (N, H, H, Cx), device=dev).to(torch.bfloat16)
        ybits = torch.randint(0, 256, (N, H, H, Cx // 8), device=dev, dtype=torch.uint8)
        mean, invstd = (torch.randint(-4, 5, (Cx,), device=dev) * 0.25).float(), (torch.randint(1, 5, (Cx,), device=dev) * 0.5).float()
        dx, part = ops.conv2d_dgrad_bn(dy, w, (N, H, H, Cx), 1, 0, addend=ad, addend_bits=abits, bn_y=y, bn_bits=ybits, bn_mean=mean, bn_invstd=invstd)
        assert ops.last_conv_kernel().startswith(want)
        _check_dgrad_bn(dev, dx, part, dy, w, ad, abits, y, ybits, mean, invstd, 2, True)
    x = torch.randint(-2, 3, (256, 7, 7, 512), device=dev).to(torch.bfloat16)
    w = torch.randint(-2, 3, (2048, 1, 1, 512), device=dev).to(torch.bfloat16)
    assert torch.equal(ops.conv2d_fwd(x, w, 1, 0, stats=True)[0], _ref(x, w, 1).to(torch.bfloat16)) and ops.last_conv_kernel() == "pk_k512_n2048_w196_s1"


def test_every_resident_weight_family_where_the_older_kernels_also_serve(dev, monkeypatch):
    """MI355_PO=2 sends the launches the default rule leaves to pk (layer 4's conv3 forward) to the resident-weight kernels, and layer 3's
    conv3 forward takes them by default: same results"""
    from sota_imagenet_amd import ops

    monkeypatch.setenv("MI355_PO", "2")
    torch.manual_seed(10)
    for (N, H, Cin, Cout, fam) in ((256, 14, 256, 1024, "po_k256_b256"), (256, 7, 512, 2048, "po_k512_b128")):
        x = torch.randint(-2, 3, (N, H, H, Cin), device=dev).to(torch.bfloat16)
        w = torch.randint(-2, 3, (Cout, 1, 1, Cin), device=dev).to(torch.bfloat16)
        y, part = ops.conv2d_fwd(x, w, 1, 0, stats=True)
        assert ops.last_conv_kernel() == fam + "_s1_a0"
        ref = _ref(x, w, 1).to(torch.bfloat16)
        assert torch.equal(y, ref)
        s2 = (ref.float() ** 2).sum(dim=(0, 1, 2)).double()
        assert (part[:, 1].double().sum(0) - s2).abs().max() <= 1e-6 * s2.abs().max()


def test_a_switch_outside_its_domain_fails_the_launch(dev, monkeypatch):
    from sota_imagenet_amd import native, ops

    x = torch.zeros(1, 8, 8, 64, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(64, 1, 1, 64, device=dev, dtype=torch.bfloat16)
    monkeypatch.setenv("MI355_DCONV", "off")
    assert native.lib().mi355_reload_knobs() != 0 and "MI355_DCONV=off" in native.last_error()
    with pytest.raises(RuntimeError, match="MI355_DCONV=off"):
        ops.conv2d_fwd(x, w, 1, 0)
    monkeypatch.delenv("MI355_DCONV")
    ops.conv2d_fwd(x, w, 1, 0)


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
             (5, 56, 64, 64), (64, 112, 64, 64),
             # 160 / 320 px (wg3_l{1,2,3}a, wg3_l{1,2,3,4}b)
             (24, 40, 64, 64), (40, 20, 128, 128), (100, 10, 256, 256), (6, 80, 64, 64), (12, 40, 128, 128), (36, 20, 256, 256), (100, 10, 512, 512),
             # BResNet-50's striding blocks (wg3_v{2,3,4})
             (9, 56, 128, 128), (20, 28, 256, 256), (256, 14, 512, 512)]


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
    assert ops.last_conv_kernel().startswith("wg3_"), ops.last_conv_kernel()
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
