#!/bin/bash
# GPU call 5: bn2 + ReLU in conv3's operand path (po_*_bn): parity, executor tests, serial traces and in-step A/B of MI355_PO_BN
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dconv_gpu.py -x -q -k "bn1 or bn2" > $O/r06e_pytest_bnin.txt 2>&1; tail -3 $O/r06e_pytest_bnin.txt
timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py -x -q -k "baseline_batch or segment_by_segment or teacher_forced_layers" > $O/r06e_pytest_exec.txt 2>&1; tail -4 $O/r06e_pytest_exec.txt
S="python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
for v in 0 1; do
  rm -rf $O/r06e_trace_bn$v
  MI355_WGRAD_STREAM=0 MI355_PO_BN=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06e_trace_bn$v -- $S > $O/r06e_bench_bn$v.json 2> $O/r06e_trace_bn$v.err
  python tools/trace_layers.py $O/r06e_trace_bn$v > $O/r06e_conv_per_layer_bn$v.txt
  python tools/timeline.py $O/r06e_trace_bn$v > $O/r06e_timeline_bn$v.txt
  rm -rf $O/r06e_trace_bn$v
done
paste <(grep "c3 " $O/r06e_conv_per_layer_bn0.txt | grep -v "\.w\|\.d") <(grep "c3 " $O/r06e_conv_per_layer_bn1.txt | grep -v "\.w\|\.d" | awk '{print $3, $4, $NF}')
head -1 $O/r06e_timeline_bn0.txt; head -1 $O/r06e_timeline_bn1.txt
grep "bn_apply" $O/r06e_timeline_bn0.txt $O/r06e_timeline_bn1.txt
timeout -k 10 500 bash tools/ab_env.sh MI355_PO_BN 0 1 resnet50 3 > $O/r06e_ab_po_bn.txt 2>&1; cat $O/r06e_ab_po_bn.txt
