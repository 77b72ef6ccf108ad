#!/bin/bash
# GPU call 4: bn-in kernels after the VALU diet: parity, then serial traces with MI355_DCONV_BN=0 / 1
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dconv_gpu.py -x -q -k "bn1" > $O/r06d_pytest_bnin.txt 2>&1; tail -3 $O/r06d_pytest_bnin.txt
export MI355_WGRAD_STREAM=0
S="python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
for v in 0 1; do
  export MI355_DCONV_BN=$v
  rm -rf $O/r06d_trace_bn$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06d_trace_bn$v -- $S > $O/r06d_bench_bn$v.json 2> $O/r06d_trace_bn$v.err
  python tools/trace_layers.py $O/r06d_trace_bn$v > $O/r06d_conv_per_layer_bn$v.txt
  python tools/timeline.py $O/r06d_trace_bn$v > $O/r06d_timeline_bn$v.txt
  rm -rf $O/r06d_trace_bn$v
done
paste <(grep "c2 " $O/r06d_conv_per_layer_bn0.txt | grep -v "\.w\|\.d") <(grep "c2 " $O/r06d_conv_per_layer_bn1.txt | grep -v "\.w\|\.d" | awk '{print $3, $4, $NF}')
head -1 $O/r06d_timeline_bn0.txt; head -1 $O/r06d_timeline_bn1.txt
grep "bn_apply" $O/r06d_timeline_bn0.txt $O/r06d_timeline_bn1.txt
