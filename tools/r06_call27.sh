#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dconv_gpu.py tests/test_fp8_gpu.py -x -q > $O/r06y_tests.log 2>&1
echo "exit $?"; tail -4 $O/r06y_tests.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for r in 1 2 3; do
  MI355RN_LIB=$PWD/sota_imagenet_amd/lib/libmi355rn_old.so timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('old', r['ms_per_step'])"
  timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('new', r['ms_per_step'])"
done
S="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary"
for v in old new; do
  rm -rf $O/r06y_$v
  if [ $v = old ]; then export MI355RN_LIB=$PWD/sota_imagenet_amd/lib/libmi355rn_old.so; else unset MI355RN_LIB; fi
  MI355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06y_$v -- $S > /dev/null 2> $O/r06y_$v.err
  f=$(ls $O/r06y_$v/*/*_kernel_stats.csv | head -1); echo "== $v"; grep "dconv_l1" $f | awk -F, '{printf "%s calls %s avg %.1f us\n", $1, $2, $4/1000}'
  rm -rf $O/r06y_$v
done
