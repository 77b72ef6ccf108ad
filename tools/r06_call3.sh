#!/bin/bash
# GPU call 3 of round 6: MFMA shapes with an unrolled loop; serial traces with MI355_DCONV_BN=0 / 1 (per-layer conv times, per-symbol totals)
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_shape_random.hip -o /tmp/mfma_shape_random 2>/dev/null && timeout -k 10 120 /tmp/mfma_shape_random > $O/r06_mfma_shape_random_data.txt 2>&1
cat $O/r06_mfma_shape_random_data.txt
export MI355_WGRAD_STREAM=0
S="python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
for v in 0 1; do
  export MI355_DCONV_BN=$v
  rm -rf $O/r06c_trace_bn$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06c_trace_bn$v -- $S > $O/r06c_bench_bn$v.json 2> $O/r06c_trace_bn$v.err
  python tools/trace_layers.py $O/r06c_trace_bn$v > $O/r06c_conv_per_layer_bn$v.txt
  python tools/timeline.py $O/r06c_trace_bn$v > $O/r06c_timeline_bn$v.txt
  find $O/r06c_trace_bn$v -name "*_kernel_trace.csv" -size +8M -delete
done
paste <(grep "c2 " $O/r06c_conv_per_layer_bn0.txt | grep -v "\.w\|\.d") <(grep "c2 " $O/r06c_conv_per_layer_bn1.txt | grep -v "\.w\|\.d" | awk '{print $3, $4, $NF}')
head -8 $O/r06c_timeline_bn0.txt; head -8 $O/r06c_timeline_bn1.txt
grep "bn_apply" $O/r06c_timeline_bn0.txt $O/r06c_timeline_bn1.txt
