"""Compile-time invariants of the conv kernels, checked on the gfx950 assembly hipcc produces (no GPU needed).

* M0: the LDS-DMA helper (csrc/lds_dma.h, blds16) writes M0 without saving it and declares it clobbered.  That is only
  sound while the compiler itself makes no use of M0 in these kernels (it would not re-materialise a value it believes
  live), so every `m0` in the assembly must be one of the helper's own `s_mov_b32 m0, s<N>` lines.
* No VGPR spills in the kernels the training step actually launches (a spill in the k-loop costs more than any of the
  tuning recorded in DESIGN.md bought); the 256x256 tile with the BN-backward epilogue is known to spill and is never
  launched (conv_igemm.hip: launch_igemm).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sota_imagenet_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(tmp_path, name):
    out = tmp_path / (name + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", os.path.join(CSRC, name + ".hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name", ["conv_igemm", "conv_wgrad"])
def test_m0_is_only_touched_by_the_dma_helper_and_nothing_spills(tmp_path, name):
    s = _asm(tmp_path, name)
    for line in s.splitlines():
        code = line.split(";")[0]
        if re.search(r"\bm0\b", code):
            assert re.match(r"\s*s_mov_b32 m0, s\d+\s*$", code), f"unexpected use of M0: {line.strip()}"
    kernels = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", s)
    assert kernels, "no kernel metadata found"
    for kname, spills in kernels:
        if "Li256ELi256ELi2E" in kname:  # 256x256 + BN-backward epilogue: instantiated, never launched
            continue
        assert int(spills) == 0, f"{kname}: {spills} spilled VGPRs"
