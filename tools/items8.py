"""Item-level timeline of the 8-wave igemm kernel (build with -DMI355_ITEMSTAMP -> sota_imagenet_amd/lib/variant_items8.so,
run with MI355RN_LIB pointing at it): per workgroup and item, k-loop start / k-loop end / epilogue issued / stores retired
on the 100 MHz realtime clock.   python tools/items8.py <tile> H Cin Cout K"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sota_imagenet_amd import native, ops

tile = sys.argv[1]
H, Cin, Cout, K = (int(v) for v in sys.argv[2:6])
N = 256
os.environ["MI355_IGEMM8"] = tile
xs = [torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16) for _ in range(4)]
w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
L = native.lib()
L.mi355_debug_items8.argtypes = [ctypes.c_void_p, ctypes.c_int]
for i in range(4):
    y = ops.conv2d_fwd(xs[i], w, 1, K // 2)
torch.cuda.synchronize()
assert L.mi355_debug_items8_clear() == 0
y = ops.conv2d_fwd(xs[0], w, 1, K // 2)
torch.cuda.synchronize()
buf = np.zeros(512 * 16 * 4, dtype=np.uint64)
assert L.mi355_debug_items8(buf.ctypes.data, buf.size) == 0
s = buf.reshape(512, 16, 4).astype(np.int64)
t0 = s[s > 0].min()
wgs = int((s[:, 0, 0] > 0).sum())
print(f"{tile} H{H} {Cin}->{Cout} k{K}: {wgs} workgroups; all times in us since the first stamp")
rel = (s - t0) / 100.0
for wg in (0, 1, 8, wgs // 2, wgs - 1):
    row = []
    for it in range(16):
        if s[wg, it, 0] == 0:
            break
        a, b, c, d = rel[wg, it]
        row.append(f"[{a:6.2f} k {b - a:5.2f} e {c - b:5.2f} r {d - c:5.2f}]")
    print(f"wg {wg:3d}: " + " ".join(row))
v = s[:wgs]
m = v[:, :, 0] > 0
kl = ((v[:, :, 1] - v[:, :, 0]) / 100.0)[m]
ep = ((v[:, :, 2] - v[:, :, 1]) / 100.0)[m]
rt = ((v[:, :, 3] - v[:, :, 2]) / 100.0)[m]
print(f"items {int(m.sum())}: k-loop mean {kl.mean():.2f} us, epilogue issue {ep.mean():.2f} us, store retire {rt.mean():.2f} us; first start {rel[:wgs, 0, 0].min():.2f}..{rel[:wgs, 0, 0].max():.2f}, last end {rel[:wgs][m][:, 3].max():.2f} us")
