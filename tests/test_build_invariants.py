"""Compile-time invariants of the conv kernels, checked on the gfx950 assembly hipcc produces (no GPU needed).

* M0: the LDS-DMA helper (csrc/lds_dma.h, blds16) writes M0 without saving it and declares it clobbered.  That is only
  sound while the compiler itself makes no use of M0 in these kernels (it would not re-materialise a value it believes
  live), so every `m0` in the assembly must be one of the helper's own `s_mov_b32 m0, s<N>` lines.
* No VGPR spills in the kernels the training step actually launches (a spill in the k-loop costs more than any of the
  tuning recorded in DESIGN.md bought); the 256x256 tile with the BN-backward epilogue is known to spill and is never
  launched (conv_igemm.hip: launch_igemm).
* The K = 128 / K = 64 f8f6f4 MFMAs of the e4m3 step are inline assembly (conv_igemm8.hip, conv_wgrad.hip): LLVM's hazard recognizer does not see into
  it, so nothing but another MFMA may touch an MFMA's destination registers within the instruction's passes + 3 wait states.
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


def _regs(tok):
    """register indices named by an operand token: v12 / a[4:7] / v[100:103] -> {('v', 12)} ..."""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name", ["conv_igemm8", "conv_wgrad"])
def test_nothing_touches_an_inline_asm_mfma_destination_inside_its_result_latency(tmp_path, name):
    """v_mfma_f32_16x16x128_f8f6f4 (8 passes) / v_mfma_f32_32x32x64_f8f6f4 (16 passes) are issued as inline asm with "+v" accumulators: hipcc's hazard
    recognizer cannot place the s_nops a VALU / accvgpr access of the destination would need.  Walk the disassembly: after each of them, for passes + 3
    wait states (s_nop n counts n + 1, every other instruction 1), no instruction other than an MFMA may name one of its destination registers."""
    s = _asm(tmp_path, name)
    lines = [l.split(";")[0].strip() for l in s.splitlines()]
    lines = [l for l in lines if l and not l.startswith((".", "//")) and not l.endswith(":")]
    need = {"v_mfma_f32_16x16x128_f8f6f4": 8 + 3, "v_mfma_f32_32x32x64_f8f6f4": 16 + 3}
    seen = 0
    for i, l in enumerate(lines):
        op = l.split()[0]
        if op not in need:
            continue
        seen += 1
        dst = _regs(l.split()[1].rstrip(","))
        assert dst, l
        left, j = need[op], i + 1
        while left > 0 and j < len(lines):
            o = lines[j].split()
            if o[0] == "s_nop":
                left -= int(o[1], 0) + 1
            else:
                left -= 1
                if not o[0].startswith("v_mfma") and not o[0].startswith("s_"):
                    touched = set()
                    for tok in re.findall(r"[va]\[\d+:\d+\]|\b[va]\d+\b", lines[j]):
                        touched |= _regs(tok)
                    assert not (touched & dst), f"{name}: `{lines[j]}` touches the destination of `{l}` {need[op] - left} wait states behind it"
            if o[0] in ("s_endpgm", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz", "s_setpc_b64"):
                break   # (control flow: the window is not followed across it; the epilogues behind the k-loop start with s_nop)
            j += 1
    assert seen > 0, "no f8f6f4 MFMA found: the check is vacuous"
