"""queue_busy.py <trace dir> — last training step of a rocprofv3 kernel trace: busy time per HIP queue (the caller's stream / the weight-gradient side
stream), the time only one of them runs, and the last kernels of each: which stream is the step's critical path"""
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd" in r["Kernel_Name"]]
step = rows[sgd[-2] + 1: sgd[-1] + 1]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
busy = collections.defaultdict(float); last = {}
for r in step:
    q = r.get("Queue_Id", "?")
    busy[q] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    last[q] = (int(r["End_Timestamp"]) - t0) / 1e6
print("step wall %.3f ms" % ((t1 - t0) / 1e6))
for q in busy:
    print("queue %s: busy %.3f ms, last kernel ends at %.3f ms, %d kernels" % (q, busy[q], last[q], sum(1 for r in step if r.get("Queue_Id", "?") == q)))
# idle gaps of each queue > 20 us
for q in busy:
    ks = [r for r in step if r.get("Queue_Id", "?") == q]
    gaps = [((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, (int(a["End_Timestamp"]) - t0) / 1e6) for a, b in zip(ks, ks[1:])]
    big = sorted(gaps, reverse=True)[:8]
    print("queue %s: idle gaps total %.3f ms; largest (us @ ms): %s" % (q, sum(g for g, _ in gaps if g > 0) / 1e3, ", ".join("%.0f@%.2f" % g for g in big)))
