#!/bin/bash
# GPU call 6: the stem pool on packed keys: bitwise test, executor switch test, per-kernel time, in-step A/B
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_resnet_gpu.py -x -q -k "packed_keys or executor_switch or teacher_forced_layers or side_stream" > $O/r06f_pytest.txt 2>&1; tail -4 $O/r06f_pytest.txt
S="python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16"
for v in 0 1; do
  rm -rf $O/r06f_trace$v
  MI355_WGRAD_STREAM=0 MI355_POOL_KEYS=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06f_trace$v -- $S > $O/r06f_bench$v.json 2> $O/r06f_trace$v.err
  python tools/timeline.py $O/r06f_trace$v > $O/r06f_timeline$v.txt
  rm -rf $O/r06f_trace$v
  grep "maxpool\|step wall" $O/r06f_timeline$v.txt
done
timeout -k 10 500 bash tools/ab_env.sh MI355_POOL_KEYS 0 1 resnet50 3 > $O/r06f_ab_pool_keys.txt 2>&1; cat $O/r06f_ab_pool_keys.txt
