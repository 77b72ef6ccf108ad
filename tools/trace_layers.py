"""Per-launch view of the conv kernels in a rocprofv3 kernel trace: maps dispatches of the LAST step to layers
(by launch order) and prints achieved TFLOP/s per launch."""
import csv, glob, sys
sys.path.insert(0, '.')
d = sys.argv[1]
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))  # valid for single-stream runs (MI355_WGRAD_STREAM=0)
conv = [r for r in rows if any(k in r["Kernel_Name"] for k in ("igemm_kernel", "igemm8_kernel", "wgrad_kernel", "conv3_kernel", "stem_direct_kernel", "dconv_", "pw_k", "pk_k", "po_k", "wg3_", "wg1_"))]
# expected launch sequence of one training step (see csrc/resnet_exec.cpp)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = int(sys.argv[3]) if len(sys.argv) > 3 else 224  # image size
seq = []  # (kind, name, flops)
def fl(Ho, Cin, Cout, K): return 2.0 * N * Ho * Ho * Cout * Cin * K * K
blocks = []
h, cin = S // 4, 64
for st, (nb, p) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
    for i in range(nb):
        s = 2 if (i == 0 and st > 0) else 1
        ho = h // s
        blocks.append((f"l{st+1}.{i}", h, ho, cin, p, s, i == 0))
        h, cin = ho, 4 * p
seq.append(("igemm", "stem", fl(S // 2, 3, 64, 7)))
for name, hin, ho, ci, p, s, ds in blocks:
    if ds: seq.append(("igemm", name + ".ds", fl(ho, ci, 4 * p, 1)))
    seq += [("igemm", name + ".c1", fl(hin, ci, p, 1)), ("igemm", name + ".c2", fl(ho, p, p, 3)), ("igemm", name + ".c3", fl(ho, p, 4 * p, 1))]
# (the head's forward / input-gradient GEMMs run on fc_kernel since round 2: not conv launches)
seq += [("wgrad", "fc.w", 2.0 * N * 2048 * 1000)]
for name, hin, ho, ci, p, s, ds in reversed(blocks):
    seq.append(("wgrad", name + ".c3.w", fl(ho, p, 4 * p, 1)))
    if ds: seq += [("igemm", name + ".ds.d", fl(ho, ci, 4 * p, 1)), ("wgrad", name + ".ds.w", fl(ho, ci, 4 * p, 1))]
    seq += [("igemm", name + ".c3.d", fl(ho, p, 4 * p, 1)),
            ("wgrad", name + ".c2.w", fl(ho, p, p, 3)), ("igemm", name + ".c2.d", fl(ho, p, p, 3)),
            ("wgrad", name + ".c1.w", fl(hin, ci, p, 1)), ("igemm", name + ".c1.d", fl(hin, ci, p, 1))]
seq.append(("wgrad", "stem.w", fl(S // 2, 3, 64, 7)))
n = len(seq)
last = conv[-n:]
tot = 0
for (kind, name, flops), r in zip(seq, last):
    assert kind in r["Kernel_Name"] or (kind == "igemm" and any(k in r["Kernel_Name"] for k in ("conv3_kernel", "stem_direct_kernel", "dconv_", "pw_k", "pk_k", "po_k"))) or (kind == "wgrad" and r["Kernel_Name"].startswith(("wg3_", "wg1_"))), (kind, name, r["Kernel_Name"][:60])
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    kn = r["Kernel_Name"]
    tag = kn if kn.startswith(("dconv_", "pw_k", "pk_k", "po_k", "wg3_", "wg1_")) else ("igemm8" if "igemm8" in kn else "")
    print(f"{name:12s} {kind:5s} {us:9.1f} us  {flops/us/1e6:8.1f} TF/s  grid {grid}  {tag}")
print("total conv us", tot)
