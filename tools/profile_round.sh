#!/bin/bash
# On the GPU box: the rocprofv3 evidence of one round -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/ afterwards).
#   kernel stats (bf16 + fp32, default overlapped run), per-layer trace + timeline (serial and overlapped), PMC traffic (serial)
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?not on a gpurun box (GRAFT_REPO_ROOT is unset)}"
mkdir -p $OUT
B="python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bf16 -- $B --dtype bf16 > $OUT/bench_bf16.jsonl 2> $OUT/bench_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fp32 -- $B --dtype fp32 --steps 6 > $OUT/bench_fp32.jsonl 2> $OUT/bench_fp32.err
export MI355_WGRAD_STREAM=0
S="python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"  # >= 10 steps: the host has to fill the launch queue before a step is in steady state
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_serial -- $S > /dev/null 2> $OUT/trace_serial.err
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $S > /dev/null 2> $OUT/pmc_$c.err; done
unset MI355_WGRAD_STREAM
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_overlapped -- $S > /dev/null 2> $OUT/trace_overlapped.err
python tools/trace_layers.py $OUT/trace_serial > $OUT/conv_per_layer_bf16_serial.txt
python tools/timeline.py $OUT/trace_serial > $OUT/timeline_bf16_serial.txt
python tools/timeline.py $OUT/trace_overlapped > $OUT/timeline_bf16_overlapped.txt
python tools/pmc_traffic.py $OUT bf16 > $OUT/pmc_traffic_bf16.json
python tools/pmc_layers.py $OUT > $OUT/pmc_per_conv_launch_bf16_serial.txt
# keep the merge small: the raw traces are tens of MB
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*_counter_collection.csv" -size +8M -delete
ls -la $OUT
