"""per-kernel-symbol median / min duration from a rocprofv3 --kernel-trace csv dir: name filter as argv[2] (substring)"""
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if flt in r["Kernel_Name"]:
        d[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:100s} n={len(v):4d} median {statistics.median(v):8.1f} us  min {min(v):8.1f} us")
