"""Host-side (no GPU) checks of the model plugin: torchvision names/shapes/order, parameter count, flat aliasing."""
import torch

from oracle.resnet50_ref import ResNet50Ref
from sota_imagenet_amd.models import resnet50


def test_state_dict_matches_torchvision_layout():
    m = resnet50()
    ref = ResNet50Ref()
    sd, rsd = m.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(rsd[k].shape), k
    # "25.56M" — configs/hydra_exp/1.r50_baseline.yaml:11
    assert sum(p.numel() for p in m.parameters()) == 25_557_032
    assert len(list(m.parameters())) == 161


def test_parameters_alias_flat_buffer_and_roundtrip():
    m = resnet50()
    ref = ResNet50Ref()
    m.load_state_dict(ref.state_dict(), strict=False)  # train.py:101 uses strict=False
    for k, v in m.state_dict().items():
        assert torch.equal(v.float(), ref.state_dict()[k].float()), k
    base = m.flat_params.data_ptr()
    n = m.flat_params.numel() * 4
    for p in m.parameters():
        assert base <= p.data_ptr() < base + n
        assert p.grad is not None and p.grad.shape == p.shape
    # conv weights are KRSC in memory (channels_last strides)
    w = m.layer1[0].conv2.weight if hasattr(m, "layer1") and isinstance(m.layer1, list) else dict(m.named_parameters())["layer1.0.conv2.weight"]
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    # writes through a parameter land in the flat array
    with torch.no_grad():
        dict(m.named_parameters())["fc.bias"].fill_(3.0)
    assert (m.flat_params == 3.0).sum().item() == 1000


def test_grad_segments_cover_flat_array_in_backward_order():
    m = resnet50()
    segs = m.grad_segments
    assert len(segs) == 18 and segs[0][0] == 0
    for (b0, e0), (b1, e1) in zip(segs, segs[1:]):
        assert e0 == b1 and e0 > b0
    assert segs[-1][1] == m.flat_grads.numel()


def test_bn_leaves_expose_momentum_for_patch_bn_mom():
    m = resnet50()
    n = 0
    for mod in m.modules():  # what pt.utils.misc.patch_bn_mom (train.py:76) walks
        if hasattr(mod, "momentum") and hasattr(mod, "running_mean"):
            mod.momentum = 0.05
            n += 1
    assert n == 53 and abs(m.bn_momentum() - 0.05) < 1e-12


def test_cpu_forward_fails_loudly():
    import pytest

    m = resnet50()
    with pytest.raises(RuntimeError, match="no CPU fallback|CUDA"):
        m.eval()(torch.zeros(1, 3, 32, 32))
    with pytest.raises((RuntimeError, ValueError)):
        m.train()(torch.zeros(1, 3, 32, 32))


def test_fp8_model_keeps_the_bf16_parameter_table():
    """resnet50(dtype="fp8") (BASELINE configs[4]): a layout-only fp8 ctx plans its e4m3 twins without a GPU and exposes the same
    flat parameter table, segments and FLOPs as the bf16 model (the twins are workspace, not parameters)."""
    from sota_imagenet_amd.models import resnet50

    a, b = resnet50(dtype="fp8"), resnet50(dtype="bf16")
    assert a.fp8 and not b.fp8 and a.compute_dtype == b.compute_dtype
    assert [(n, tuple(p.shape)) for n, p in a.named_parameters()] == [(n, tuple(p.shape)) for n, p in b.named_parameters()]
    assert a.grad_segments == b.grad_segments and a.flops(256, 224, 224) == b.flops(256, 224, 224)
    assert a.bucket_plan(32.0) == b.bucket_plan(32.0)


def test_bresnet50_executor_facade_layout_matches_the_per_op_graph():
    """bresnet.BResNet50 (views into the static executor's flat arrays, layout-only context: no GPU) and bresnet.BResNet50Graph (separate
    nn.Parameters) expose the same pytorch_tools names, shapes, registration order and initial values; no CPU compute path exists."""
    import pytest

    from sota_imagenet_amd.bresnet import BResNet50, BResNet50Graph

    m, g = BResNet50(weight_standardization=True), BResNet50Graph(weight_standardization=True)
    a, b = m.state_dict(), g.state_dict()
    assert list(a) == list(b) and len(a) == 348
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k].float(), b[k].float()), k
    assert [(n, tuple(p.shape)) for n, p in m.named_parameters()] == [(n, tuple(p.shape)) for n, p in g.named_parameters()]
    assert sum(p.numel() for p in m.parameters()) == 25576312
    w = dict(m.named_parameters())["layer3.0.conv2.weight"]
    assert w.shape == (256, 256, 3, 3) and w.stride() == (2304, 1, 768, 256)  # OIHW logical over [Cout][KH][KW][Cin] memory
    assert w.untyped_storage().data_ptr() == m.flat_params.untyped_storage().data_ptr() and w.grad.shape == w.shape
    segs = m.grad_segments  # head, 16 bottlenecks last to first, stem: descending through the forward-ordered flat array, tiling it
    assert len(segs) == 18 and segs[0][1] == m.flat_params.numel() and segs[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(segs, segs[1:]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 3, 64, 64))
    with pytest.raises(ValueError):
        BResNet50(dtype="fp8")
