"""fp8 (OCP e4m3fn) operand form of the convolution (BASELINE.json configs[4]; csrc/fp8.hip + conv_igemm8.hip EB=1) on the GPU.
  * the quantiser is bit-exact against the oracle grid (oracle/ops_ref.py::quantize_e4m3, pinned to torch's cast on CPU);
  * on small-integer data every product and partial sum is exact, so the fp8 conv must equal the bf16 conv BIT FOR BIT;
  * on scaled random data: within one bf16 rounding of the fp32 conv of the dequantised operands (tolerance 2^-8 of max)."""
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn
# (N, H, W, Cin, Cout, K, stride): ragged M tiles, both tile widths (Cout % 256 == 0 -> 224x256, else 256x128), taps 1 and 9
CASES = [(2, 14, 14, 128, 256, 3, 1), (3, 7, 7, 256, 128, 3, 1), (4, 14, 14, 256, 256, 1, 1), (5, 10, 10, 128, 256, 3, 2),
         (2, 8, 8, 128, 256, 1, 2), (16, 14, 14, 256, 512, 3, 1), (40, 14, 14, 128, 128, 3, 1), (33, 7, 7, 512, 256, 1, 1),
         (1, 2, 2, 128, 128, 1, 1), (1, 2, 2, 128, 256, 3, 1)]


def test_quantize_bit_exact(dev):
    from sota_imagenet_amd import ops

    g = torch.Generator().manual_seed(1)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.cat([(torch.rand(4096, generator=g) * 2 - 1) * 600, torch.randn(4096, generator=g), torch.randn(4096, generator=g) * 0.01]).to(dtype)
        for scale in (1.0, 0.5, 3.0):
            q = ops.quantize_fp8(x.to(dev), scale)
            assert q.dtype == F8
            # the kernel multiplies in fp32, like the oracle's x.float()*scale for power-of-two and small-integer scales
            want = R.quantize_e4m3((x.float() * scale))
            assert torch.equal(q.cpu().float(), want), (dtype, scale)
    with pytest.raises(Exception):
        ops.quantize_fp8(torch.zeros(12, device=dev))  # numel % 8


@pytest.mark.parametrize("case", CASES)
def test_conv_fp8_equals_bf16_on_integers(dev, case):
    from sota_imagenet_amd import ops

    N, H, W, Cin, Cout, K, s = case
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // s + 1, (W + 2 * pad - K) // s + 1
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-2, 3, (N, H, W, Cin), generator=g).float().to(dev)
    w = torch.randint(-2, 3, (Cout, K, K, Cin), generator=g).float().to(dev)
    dy = torch.randint(-2, 3, (N, Ho, Wo, Cout), generator=g).float().to(dev)
    xq, wq, dyq = ops.quantize_fp8(x), ops.quantize_fp8(w), ops.quantize_fp8(dy)
    assert torch.equal(xq.float(), x)
    y8 = ops.conv2d_fwd_fp8(xq, wq, s, pad)
    y16 = ops.conv2d_fwd(x.bfloat16(), w.bfloat16(), s, pad)
    assert y8.dtype == torch.bfloat16 and torch.equal(y8, y16)
    assert torch.equal(y8.float().cpu(), R.conv2d_fwd(x.cpu(), w.cpu(), s, pad).bfloat16().float())
    dx8 = ops.conv2d_dgrad_fp8(dyq, wq, (N, H, W, Cin), s, pad)
    dx16 = ops.conv2d_dgrad(dy.bfloat16(), w.bfloat16(), (N, H, W, Cin), s, pad)
    assert torch.equal(dx8, dx16)
    # output scale is applied in fp32 before the single bf16 rounding
    y8h = ops.conv2d_fwd_fp8(xq, wq, s, pad, oscale=0.5)
    assert torch.equal(y8h.float(), (y16.float() * 0.5))  # small integers: halving stays exact in bf16


@pytest.mark.parametrize("case", [(8, 14, 14, 256, 256, 3, 1), (8, 28, 28, 256, 256, 3, 2), (8, 7, 7, 2048, 512, 1, 1), (8, 14, 14, 256, 1024, 1, 1)])
def test_conv_fp8_scaled_random(dev, case):
    from sota_imagenet_amd import ops

    N, H, W, Cin, Cout, K, s = case
    pad = K // 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, K, K, Cin), generator=g) * 0.05
    sx, sw = 448.0 / x.abs().max().item() / 2, 448.0 / w.abs().max().item() / 2  # amax scaling with one bit of headroom
    xq, wq = ops.quantize_fp8(x.to(dev), sx), ops.quantize_fp8(w.to(dev), sw)
    xv, wv = R.quantize_e4m3(x, sx), R.quantize_e4m3(w, sw)
    assert torch.equal(xq.cpu().float(), xv) and torch.equal(wq.cpu().float(), wv)
    osc = 1.0 / (sx * sw)
    y = ops.conv2d_fwd_fp8(xq, wq, s, pad, oscale=osc).float().cpu()
    ref = R.conv2d_fwd_fp8(xv, wv, s, pad, osc)
    assert (y - ref).abs().max() <= ref.abs().max() * 2.0 ** -8
    # and the quantisation itself is a small perturbation of the fp32 conv (sanity of the scaling recipe, not a parity bar)
    full = R.conv2d_fwd(x, w, s, pad)
    assert ((ref - full).norm() / full.norm()).item() < 0.06
    Ho, Wo = y.shape[1:3]
    dy = torch.randn((N, Ho, Wo, Cout), generator=g)
    sd = 448.0 / dy.abs().max().item() / 2
    dyq, dyv = ops.quantize_fp8(dy.to(dev), sd), R.quantize_e4m3(dy, sd)
    dx = ops.conv2d_dgrad_fp8(dyq, wq, (N, H, W, Cin), s, pad, oscale=1.0 / (sd * sw)).float().cpu()
    dref = R.conv2d_dgrad_fp8(dyv, wv, (N, H, W, Cin), s, pad, 1.0 / (sd * sw))
    assert (dx - dref).abs().max() <= dref.abs().max() * 2.0 ** -8


@pytest.mark.parametrize("case", [(4, 14, 14, 128, 256, 3, 1), (3, 7, 7, 256, 128, 1, 1), (5, 10, 10, 128, 128, 3, 2), (2, 8, 8, 256, 256, 1, 2),
                                  (64, 14, 14, 256, 256, 3, 1), (33, 7, 7, 512, 128, 1, 1)])
def test_wgrad_fp8(dev, case):
    """weight gradient on e4m3 operands (transposed byte reads): on small integers every product and partial sum is exact, so it must
    equal the bf16 weight gradient of the same data BIT FOR BIT; the output scale is one fp32 multiply."""
    from sota_imagenet_amd import ops

    N, H, W, Cin, Cout, K, s = case
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // s + 1, (W + 2 * pad - K) // s + 1
    g = torch.Generator().manual_seed(13)
    x = torch.randint(-2, 3, (N, H, W, Cin), generator=g).float().to(dev)
    dy = torch.randint(-2, 3, (N, Ho, Wo, Cout), generator=g).float().to(dev)
    dw8 = ops.conv2d_wgrad_fp8(ops.quantize_fp8(dy), ops.quantize_fp8(x), K, K, s, pad)
    dw16 = ops.conv2d_wgrad(dy.bfloat16(), x.bfloat16(), K, K, s, pad)
    assert torch.equal(dw8, dw16)
    _, dwref = R.conv2d_bwd(x.cpu(), torch.zeros(Cout, K, K, Cin), dy.cpu(), s, pad)
    assert torch.equal(dw8.cpu(), dwref)
    assert torch.equal(ops.conv2d_wgrad_fp8(ops.quantize_fp8(dy), ops.quantize_fp8(x), K, K, s, pad, oscale=0.25), dw16 * 0.25)


def test_conv_fp8_rejects_unsupported_shapes(dev):
    from sota_imagenet_amd import ops

    xq = torch.zeros((2, 8, 8, 64), dtype=torch.uint8, device=dev).view(F8)
    wq = torch.zeros((128, 3, 3, 64), dtype=torch.uint8, device=dev).view(F8)
    with pytest.raises(Exception, match="multiples of 128"):
        ops.conv2d_fwd_fp8(xq, wq, 1, 1)
    xq = torch.zeros((2, 1, 1, 128), dtype=torch.uint8, device=dev).view(F8)  # one-pixel rows: not a shape of this network
    wq = torch.zeros((128, 1, 1, 128), dtype=torch.uint8, device=dev).view(F8)
    with pytest.raises(Exception, match="at least 2"):
        ops.conv2d_fwd_fp8(xq, wq, 1, 0)


# (N, H, C, family): conv2 of the stride-1 bottlenecks of layers 2 - 4 at 224 px — the e4m3 step's 3x3 launches on the generated kernels (asm/dconv_gen.py
# Cfg.fp8: 128-channel chunks, v_mfma_f32_16x16x128_f8f6f4), + ragged batches
Q3X3 = [(256, 28, 128, "dconv_l2"), (256, 14, 256, "dconv_l3"), (256, 7, 512, "dconv_l4"), (5, 14, 256, "dconv_l3"), (6, 7, 512, "dconv_l4")]


@pytest.mark.parametrize("N,H,C,fam", Q3X3)
def test_generated_e4m3_3x3_kernels_equal_the_bf16_kernels_on_integers(dev, N, H, C, fam, monkeypatch):
    """forward and data gradient on e4m3 operands through mi355_conv2d_fwd_fp8 / _dgrad_fp8: small integers are exact in e4m3, bf16 and every partial sum,
    so the generated e4m3 kernel, the generated bf16 kernel and (MI355_DCONV_FP8=0) the 8-wave e4m3 kernel must agree bit for bit; the launch is
    asserted by name"""
    from sota_imagenet_amd import native, ops

    g = torch.Generator().manual_seed(31)
    x = torch.randint(-2, 3, (N, H, H, C), generator=g).float().to(dev)
    w = torch.randint(-2, 3, (C, 3, 3, C), generator=g).float().to(dev)
    dy = torch.randint(-2, 3, (N, H, H, C), generator=g).float().to(dev)
    xq, wq, dyq = ops.quantize_fp8(x), ops.quantize_fp8(w), ops.quantize_fp8(dy)
    y8 = ops.conv2d_fwd_fp8(xq, wq, 1, 1)
    assert ops.last_conv_kernel() == fam + "_s0_q", ops.last_conv_kernel()
    y16 = ops.conv2d_fwd(x.bfloat16(), w.bfloat16(), 1, 1)
    assert ops.last_conv_kernel() == fam + "_s0"
    assert torch.equal(y8, y16)
    dx8 = ops.conv2d_dgrad_fp8(dyq, wq, (N, H, H, C), 1, 1)
    assert ops.last_conv_kernel() == fam + "_s0_q", ops.last_conv_kernel()
    assert torch.equal(dx8, ops.conv2d_dgrad(dy.bfloat16(), w.bfloat16(), (N, H, H, C), 1, 1))
    y8h = ops.conv2d_fwd_fp8(xq, wq, 1, 1, oscale=0.25)
    assert torch.equal(y8h.float(), y16.float() * 0.25)
    monkeypatch.setenv("MI355_DCONV_FP8", "0")
    native.lib().mi355_reload_knobs()
    y8o = ops.conv2d_fwd_fp8(xq, wq, 1, 1)
    assert not ops.last_conv_kernel().startswith("dconv"), ops.last_conv_kernel()
    assert torch.equal(y8o, y8)


def test_generated_e4m3_3x3_kernel_on_scaled_random_data(dev):
    """random operands quantised with per-tensor scales: within one bf16 rounding of the fp32 convolution of the dequantised operands (the bar of
    test_conv_fp8_scaled_random), and within summation-order noise of the 8-wave e4m3 kernel"""
    from sota_imagenet_amd import ops

    torch.manual_seed(32)
    N, H, C = 32, 14, 256
    x, w = torch.randn(N, H, H, C, device=dev), torch.randn(C, 3, 3, C, device=dev) * 0.05
    sx, sw = 448.0 / (2 * x.abs().max().item()), 448.0 / (2 * w.abs().max().item())
    xq, wq = ops.quantize_fp8(x, sx), ops.quantize_fp8(w, sw)
    y8 = ops.conv2d_fwd_fp8(xq, wq, 1, 1, oscale=1.0 / (sx * sw))
    assert ops.last_conv_kernel() == "dconv_l3_s0_q"
    ref = torch.nn.functional.conv2d((xq.float() / sx).permute(0, 3, 1, 2), (wq.float() / sw).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert (y8.float() - ref).abs().max() <= 2.0 ** -8 * ref.abs().max()
