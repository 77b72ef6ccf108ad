#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --dtype fp8 --batch 512 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary"
rm -rf $O/r06l_ser $O/r06l_ovl
MI355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06l_ser -- $S > /dev/null 2> $O/r06l_ser.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06l_ovl -- $S > /dev/null 2> $O/r06l_ovl.err
n=$(python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r06l_ser/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sgd=[i for i,r in enumerate(rows) if "sgd" in r["Kernel_Name"]]
print(sgd[-1]-sgd[-2])
PY
)
python tools/stream_table.py $O/r06l_ser $O/r06l_ovl $n > $O/r06l_per_symbol_fp8_bs512.txt; head -40 $O/r06l_per_symbol_fp8_bs512.txt
rm -rf $O/r06l_ser $O/r06l_ovl
