#!/bin/bash
# On the GPU box: the per-layer conv table of a SERIAL step and the step time of the default (two-stream) step with one environment switch
# at each of its values: tools/po_ab.sh <VAR> <v1> <v2> ... -> gpurun_out/ab_<VAR>/
VAR=${1:?variable}; shift
OUT=gpurun_out/ab_$VAR
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?not on a gpurun box (GRAFT_REPO_ROOT is unset)}"
mkdir -p $OUT
S="python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
for v in "$@"; do
  export $VAR=$v
  MI355_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$v -- $S > /dev/null 2> $OUT/trace_$v.err
  python tools/trace_layers.py $OUT/trace_$v > $OUT/conv_per_layer_$v.txt 2>> $OUT/trace_$v.err
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  rm -rf $OUT/trace_$v
done
for r in 1 2; do for v in "$@"; do
  export $VAR=$v
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16 >> $OUT/bench_$v.json 2>> $OUT/bench_$v.err
done; done
for v in "$@"; do echo "$VAR=$v: $(tail -1 $OUT/conv_per_layer_$v.txt)  ms/step: $(python3 -c "
import json,sys
print(' '.join('%.3f' % json.loads(l)['ms_per_step'] for l in open('$OUT/bench_$v.json') if l.startswith('{')))")"; done | tee $OUT/summary.txt
