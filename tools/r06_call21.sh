#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary"
rm -rf $O/r06t
MI355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06t -- $S > /dev/null 2> $O/r06t.err
f=$(ls $O/r06t/*/*_kernel_trace.csv | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sgd=[i for i,r in enumerate(rows) if "sgd" in r["Kernel_Name"]]
step=rows[sgd[-2]+1:sgd[-1]+1]
for i,r in enumerate(step):
    n=r["Kernel_Name"]
    if "bn_reduce" in n:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        print("%4d %-70s %6.1f us   prev: %s | next: %s"%(i, n[:70], d, step[i-1]["Kernel_Name"][:50], step[i+1]["Kernel_Name"][:40]))
PY
rm -rf $O/r06t
