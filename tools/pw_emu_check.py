"""pw_emu_check.py — run a generated pointwise kernel (csrc/asm/pw_gen.py) in the CPU emulator (tools/gcn_emu.py) against
numpy on exact small-integer data.  Test infrastructure; used by tests/test_dconv_emu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))

import gcn_emu  # noqa: E402
import pw_gen  # noqa: E402
from dconv_emu_check import bf16_round, from_bf16_bits, to_bf16_bits  # noqa: E402


def run(name, mtiles=2, grid=2, upw=None, seed=0, check=True, **over):
    c, g, text = pw_gen.generate(name, **over)
    rng = np.random.default_rng(seed)
    M = mtiles * c.ROWS
    x = rng.integers(-2, 3, size=(M, c.K)).astype(np.float32)
    w = rng.integers(-2, 3, size=(c.N, c.K)).astype(np.float32)
    units = mtiles * c.NTN
    if upw is None:
        upw = -(-units // grid)
    mem = gcn_emu.Memory()
    a_in, a_wt = mem.alloc(to_bf16_bits(x)), mem.alloc(to_bf16_bits(w))
    out0 = np.full((M, c.N), 0x7FC0, dtype=np.uint16)
    a_out = mem.alloc(out0)
    a_stat = mem.alloc(np.full((grid, 2, c.N), np.nan, dtype=np.float32))
    fields = [("q", a_in), ("q", a_wt), ("q", a_out), ("q", a_stat)] + [("q", 0)] * 6 + [("I", units), ("I", upw), ("I", mtiles), ("I", 0)]
    fields += [("I", 0)] * 8
    fields += [("I", v) for row in pw_gen.tables(c) for v in row]
    ka = gcn_emu.pack_kernarg(fields)
    assert len(ka) == pw_gen.Gen.KA["size"], len(ka)
    a_ka = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    total = 0
    for j in range(grid):
        emu = gcn_emu.Emulator(text, mem, lds_bytes=c.LDS, check=check)
        total += emu.run_workgroup(4, a_ka, wg_id=(j, 0, 0))
    got = from_bf16_bits(mem.array(a_out, np.uint16, out0.shape)).astype(np.float64)
    refr = bf16_round((x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)).astype(np.float64)
    res = {"insts": total, "max_err": float(np.abs(got - refr).max())}
    if c.stats == 1:
        st = mem.array(a_stat, np.float32, (grid, 2, c.N)).astype(np.float64)
        res["stat_err"] = float(max(np.abs(st[:, 0].sum(0) - refr.sum(0)).max(), np.abs(st[:, 1].sum(0) - (refr ** 2).sum(0)).max()))
    return res


if __name__ == "__main__":
    import time
    t0 = time.time()
    print(run("pw_k256_n1024_s1", mtiles=2, grid=2, K=128, N=512), "%.1f s" % (time.time() - t0))
