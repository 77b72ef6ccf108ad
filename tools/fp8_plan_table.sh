#!/bin/bash
# on the GPU box: per-layer conv tables of a SERIAL bs-512 / 224 px step: bf16, fp8 by the plan, fp8 with every legal launch on e4m3 (MI355_FP8_PLAN=all)
OUT=gpurun_out/fp8plan; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p $OUT
for tag in bf16 fp8 fp8all; do
  dt=$tag; [ $tag = fp8all ] && { dt=fp8; export MI355_FP8_PLAN=all; }
  MI355_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$tag -- python3 bench.py --steps 6 --warmup 3 --batch 512 --no-cpu-baseline --no-roofline --no-secondary --dtype $dt > /dev/null 2> $OUT/$tag.err
  python tools/trace_layers.py $OUT/t_$tag > $OUT/conv_per_layer_${tag}_bs512.txt 2>> $OUT/$tag.err
  rm -rf $OUT/t_$tag
  unset MI355_FP8_PLAN
done
tail -1 $OUT/conv_per_layer_*_bs512.txt
