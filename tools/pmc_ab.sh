#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE passes of a serial bf16 step under the current environment -> gpurun_out/<tag>/pmc_per_conv_launch.txt
TAG=${1:?tag}
export TMPDIR=/tmp MI355_WGRAD_STREAM=0
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
S="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --dtype ${DTYPE:-bf16}"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $S > /dev/null 2> $OUT/pmc_$c.err; done
python tools/pmc_layers.py $OUT > $OUT/pmc_per_conv_launch.txt
python tools/pmc_traffic.py $OUT ${DTYPE:-bf16} > $OUT/pmc_traffic.json
find $OUT -name "*_kernel_trace.csv" -size +4M -delete
find $OUT -name "*_counter_collection.csv" -size +4M -delete
