#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of a serial (one-stream) step for each dtype given -> gpurun_out/<tag>/<dtype>_top.txt
TAG=${1:?tag}; shift
export TMPDIR=/tmp MI355_WGRAD_STREAM=${MI355_WGRAD_STREAM:-0}
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
for dt in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$dt -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --dtype $dt $BENCH_EXTRA > $OUT/$dt.json 2> $OUT/$dt.err
  python tools/stats_top.py $OUT/$dt 40 > $OUT/${dt}_top.txt
done
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
cat $OUT/*_top.txt
