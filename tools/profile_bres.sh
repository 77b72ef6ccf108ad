#!/bin/bash
# On the GPU box: rocprofv3 kernel stats + timeline of the BResNet-50 step (configs[3]) -> gpurun_out/prof_bres_<tag>/
TAG=${1:-r05}
OUT=gpurun_out/prof_bres_$TAG
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?not on a gpurun box (GRAFT_REPO_ROOT is unset)}"
mkdir -p $OUT
B="python3 bench.py --model bresnet50 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench.jsonl 2> $OUT/bench.err
python tools/timeline.py $OUT/stats > $OUT/timeline.txt
S="python3 bench.py --model bresnet50 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
for c in FETCH_SIZE WRITE_SIZE; do MI355_WGRAD_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $S > /dev/null 2> $OUT/pmc_$c.err; done
python tools/pmc_total.py $OUT > $OUT/pmc_traffic_bresnet50_bf16.json
find $OUT -name "*_counter_collection.csv" -size +8M -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls -la $OUT $OUT/stats/*
