#!/bin/bash
# GPU call 9: e4m3 step at bs 512, serial kernel stats with MI355_DCONV_FP8=0 / 1
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --dtype fp8 --batch 512 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary"
for v in 0 1; do
  rm -rf $O/r06i_stats$v
  MI355_WGRAD_STREAM=0 MI355_DCONV_FP8=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06i_stats$v -- $S > $O/r06i_bench$v.json 2> $O/r06i_stats$v.err
  f=$(find $O/r06i_stats$v -name "*kernel_stats.csv" | head -1)
  echo "== MI355_DCONV_FP8=$v"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "dconv" in n or "igemm8" in n:
        print("%-90s calls %5s avg %8.1f us total %8.1f ms" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  find $O/r06i_stats$v -name "*_kernel_trace.csv" -delete
done
