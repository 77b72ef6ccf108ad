"""Per-launch HBM traffic of the conv kernels: maps the igemm / wgrad dispatches of the LAST step in two rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE; serial run, MI355_WGRAD_STREAM=0) to layers by launch order (same sequence as
tools/trace_layers.py) and prints corrected HBM bytes next to the algorithmic bytes of the launch."""
import csv, glob, sys
root = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 256; ES = 2
def load(c):
    f = glob.glob(f"{root}/pmc_{c}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [r for r in rows if any(k in r["Kernel_Name"] for k in ("igemm_kernel", "igemm8_kernel", "wgrad_kernel", "stem_direct_kernel", "dconv_", "pw_k", "pk_k", "po_k", "wg3_", "wg1_"))]
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
seq = []
def el(h, c): return N * h * h * c
blocks = []; h, cin = 56, 64
for st, (nb, p) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
    for i in range(nb):
        s = 2 if (i == 0 and st > 0) else 1
        blocks.append((f"l{st+1}.{i}", h, h // s, cin, p, i == 0)); h, cin = h // s, 4 * p
seq.append(("igemm", "stem", None))
for name, hin, ho, ci, p, ds in blocks:
    if ds: seq.append(("igemm", name + ".ds", (el(hin, ci) if hin == ho else el(hin, ci) / 4) + el(ho, 4 * p)))
    seq += [("igemm", name + ".c1", el(hin, ci) + el(hin, p)), ("igemm", name + ".c2", el(hin, p) + el(ho, p)), ("igemm", name + ".c3", el(ho, p) + el(ho, 4 * p))]
seq += [("wgrad", "fc.w", None)]  # (the head's forward / input-gradient GEMMs run on fc_kernel: not conv launches)
for name, hin, ho, ci, p, ds in reversed(blocks):
    seq.append(("wgrad", name + ".c3.w", el(ho, p) + el(ho, 4 * p)))
    if ds: seq += [("igemm", name + ".ds.d", el(ho, 4 * p) + el(hin, ci)), ("wgrad", name + ".ds.w", el(ho, 4 * p) + el(hin, ci))]
    seq += [("igemm", name + ".c3.d", el(ho, 4 * p) + 2 * el(ho, p)), ("wgrad", name + ".c2.w", el(ho, p) + el(hin, p)),
            ("igemm", name + ".c2.d", el(ho, p) + 2 * el(hin, p)), ("wgrad", name + ".c1.w", el(hin, p) + el(hin, ci)),
            ("igemm", name + ".c1.d", el(hin, p) + 3 * el(hin, ci))]
seq.append(("wgrad", "stem.w", None))
n = len(seq)
for (kind, name, alg), f, w in zip(seq, fe[-n:], wr[-n:]):
    assert kind in f["Kernel_Name"] or (kind == "igemm" and any(k in f["Kernel_Name"] for k in ("stem_direct_kernel", "dconv_", "pw_k", "pk_k", "po_k"))) or (kind == "wgrad" and f["Kernel_Name"].startswith(("wg3_", "wg1_"))), (kind, name, f["Kernel_Name"][:50])
    hbm = (2 * float(f["Counter_Value"]) + float(w["Counter_Value"])) * 1024
    a = f"{alg*ES/1e6:8.1f}" if alg else "       -"
    r = f"{hbm/(alg*ES):5.2f}x" if alg else ""
    print(f"{name:12s} {kind:5s} hbm {hbm/1e6:8.1f} MB  alg {a} MB  {r}")
