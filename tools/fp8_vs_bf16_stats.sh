#!/bin/bash
# on the GPU box: rocprofv3 kernel stats of the bs-512 / 224 px step in bf16 and fp8, per symbol side by side (ms per step)
OUT=gpurun_out/fp8stats; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p $OUT
for dt in bf16 fp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$dt -- python3 bench.py --steps 10 --warmup 4 --batch 512 --no-cpu-baseline --no-roofline --no-secondary --dtype $dt > $OUT/$dt.json 2> $OUT/$dt.err
done
python3 - <<'PY'
import csv, glob, re, collections
def load(dt):
    f = glob.glob(f"gpurun_out/fp8stats/{dt}/**/*_kernel_stats.csv", recursive=True)[0]
    d = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("mi355::", "")
        n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)[:60]
        n = re.sub(r"_ZN5mi35512_GLOBAL__N_1\d+", "", n)
        d[n] += float(r["TotalDurationNs"]) / 14 / 1e6   # 14 steps incl. warm-up
    return d
b, f = load("bf16"), load("fp8")
keys = sorted(set(b) | set(f), key=lambda k: -(b.get(k, 0) + f.get(k, 0)))
print("%-62s %8s %8s" % ("kernel (ms per step, summed over both streams)", "bf16", "fp8"))
for k in keys[:45]:
    print("%-62s %8.3f %8.3f" % (k, b.get(k, 0), f.get(k, 0)))
print("%-62s %8.3f %8.3f" % ("TOTAL", sum(b.values()), sum(f.values())))
PY
