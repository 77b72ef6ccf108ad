#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_variant_gpu.py -x -q  > $O/r06x_tests.log 2>&1
echo "exit $?"; tail -3 $O/r06x_tests.log
B="python bench.py --model bresnet50 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for r in 1 2 3; do
  MI355RN_LIB=$PWD/sota_imagenet_amd/lib/libmi355rn_old.so timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('old', r['ms_per_step'])"
  timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('new', r['ms_per_step'])"
done
