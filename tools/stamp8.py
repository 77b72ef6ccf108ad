"""Where a phase of the 8-wave igemm kernel spends its cycles: runs one conv with the -DMI355_STAMP8 build
(MI355RN_LIB=sota_imagenet_amd/lib/variant_stamp8.so) and prints, per phase of the k-loop and per wave row, the average
cycles of  L = fragment reads + LDS-DMA issue + counted wait | B1 = first barrier + lgkmcnt | M = MFMA issue | B2 = second barrier."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sota_imagenet_amd import native, ops

tile = sys.argv[1] if len(sys.argv) > 1 else "256x256"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H, Cin, Cout, K = 14, 256, 256, 3
os.environ["MI355_IGEMM8"] = tile
x = torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
for _ in range(3):
    y = ops.conv2d_fwd(x, w, 1, 1)
torch.cuda.synchronize()
L = native.lib()
L.mi355_debug_stamps8.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(2 * 1024, dtype=np.uint64)
assert L.mi355_debug_stamps8(buf.ctypes.data, buf.size) == 0
for wr in range(2):
    s = buf[wr * 1024:(wr + 1) * 1024].reshape(-1, 4).astype(np.int64)
    n = int((s[:, 0] > 0).sum())
    s = s[:n]
    nph = 4 if tile.endswith("256") else 2
    print(f"wave row {wr}: {n} phases stamped; cycles per k-tile (steady state, k-tiles 4..): "
          f"{(s[nph * 20, 0] - s[nph * 4, 0]) / 16:.0f}")
    for p in range(nph):
        idx = np.arange(nph * 4 + p, n - nph, nph)
        Ls = (s[idx, 1] - s[idx, 0]).mean()
        B1 = (s[idx, 2] - s[idx, 1]).mean()
        M = (s[idx, 3] - s[idx, 2]).mean()
        B2 = (s[idx + 1, 0] - s[idx, 3]).mean()
        print(f"  phase {p + 1}: L {Ls:6.0f}  B1 {B1:6.0f}  M {M:6.0f}  B2 {B2:6.0f}   sum {Ls + B1 + M + B2:6.0f}")
