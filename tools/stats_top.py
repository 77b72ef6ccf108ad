"""top kernels of a rocprofv3 --kernel-trace --stats output dir: name, calls, total ms, avg us (names shortened)"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = re.sub(r"mi355::\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"void ", "", n)[:90]
    print(f"{n:92s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
