"""The C-ABI library loads on a GPU-less host, exports every symbol include/mi355rn.h declares, and fails loudly
(status + message, no crash, no CPU fallback) when asked to compute without a device."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from sota_imagenet_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    protos = native.parse_header()
    assert len(protos) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\s[TW]\s+(mi355_\w+)", out))
    assert set(protos) <= exported, sorted(set(protos) - exported)
    assert exported <= set(protos), f"exported but undeclared: {sorted(exported - set(protos))}"
    # no torch types leak into the signatures
    for name, (ret, params) in protos.items():
        for p in params:
            assert "Tensor" not in p and "at::" not in p and "torch" not in p, (name, p)


def test_library_is_built_for_gfx950():
    out = subprocess.run(["strings", "-n", "6", native.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_layout_only_context_and_param_table():
    L = native.lib()
    assert L.mi355_version() >= 100
    ctx = ctypes.c_void_p()
    native.check(L.mi355_resnet50_create(ctypes.byref(ctx), -1, native.F32, 256, 224, 224, 1000))
    try:
        f, t = ctypes.c_double(), ctypes.c_double()
        native.check(L.mi355_resnet50_flops(ctx, ctypes.byref(f), ctypes.byref(t)))
        # 8.178 GFLOP fwd / 24.30 GFLOP train per image
        assert abs(f.value / 256 / 1e9 - 8.178) < 2e-3 and abs(t.value / 256 / 1e9 - 24.30) < 5e-3
        assert L.mi355_resnet50_num_segments(ctx) == 18
        assert L.mi355_resnet50_num_tensors(ctx) == 161 + 106
        # layout-only contexts refuse to compute
        rc = L.mi355_resnet50_forward(ctx, None, None, 1, 0.1, None)
        assert rc != 0 and native.last_error()
        buf = (ctypes.c_float * 64)()
        rc = L.mi355_resnet50_bind(ctx, ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p))
        assert rc != 0
    finally:
        L.mi355_resnet50_destroy(ctx)


def test_bad_arguments_return_status_not_crash():
    L = native.lib()
    rc = L.mi355_resnet50_create(None, -1, 0, 1, 32, 32, 10)
    assert rc == -1 and "null" in native.last_error()
    ctx = ctypes.c_void_p()
    assert L.mi355_resnet50_create(ctypes.byref(ctx), -1, 7, 1, 32, 32, 10) == -1  # bad dtype
    assert L.mi355_resnet50_create(ctypes.byref(ctx), -1, 0, 1, 33, 32, 10) == -1  # H not a multiple of 32
    assert L.mi355_conv2d_fwd(0, None, None, None, 1, 8, 8, 60, 64, 3, 3, 1, 1, None) == -1  # Cin % 64
    assert "Cin" in native.last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the GPU-less failure mode")
def test_no_cpu_fallback_without_gpu():
    L = native.lib()
    assert L.mi355_device_count() == 0
    x = torch.zeros(64 * 4)
    rc = L.mi355_sgd_step(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.numel(), 0.1, 0.9, 0.0, 1.0, None)
    assert rc == -2 and "hip" in native.last_error().lower()  # MI355_E_HIP: the launch fails, nothing is computed
    assert torch.count_nonzero(x) == 0
    from sota_imagenet_amd import ops

    with pytest.raises(ValueError):
        ops.conv2d_fwd(torch.zeros(1, 8, 8, 64), torch.zeros(64, 1, 1, 64))
