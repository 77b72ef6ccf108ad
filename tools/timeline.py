"""Timeline view of the LAST training step in a rocprofv3 kernel trace: wall time of the step, time with no kernel
running, per-kernel-symbol busy time and how much of it ran concurrently with another kernel (side stream)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
import re
def short(n):
    m = re.search(r"(\w+_kernel)(I[^E]*E|<[^>]*>)?", n)
    return (m.group(0) if m else n)[:60]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd" in r["Kernel_Name"]]
lo, hi = sgd[-2] + 1, sgd[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = ovl = 0; depth = 0; last = t0
for t, dlt in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: ovl += t - last
    depth += dlt; last = t
print(f"step wall {(t1-t0)/1e6:.3f} ms, kernels {len(step)}, busy {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms, >=2 kernels {ovl/1e6:.3f} ms")
by = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = short(r["Kernel_Name"])
    by[n][0] += 1; by[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for n, (k, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.3f} ms {k:4d}  {n}")

for r in step:
    if "ce_row" in r["Kernel_Name"]:
        print(f"forward ends (ce_row starts) at {(int(r['Start_Timestamp'])-t0)/1e3:.1f} us")
# per-phase busy time by kernel on the main queue
if len(sys.argv) > 2:
    k = int(sys.argv[2])
    sel = step[:k] + step[-k:]
    for r in sel:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  q{r.get('Queue_Id','?')} {short(r['Kernel_Name'])}")
