#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_fp8_step_gpu.py -x -q > $O/r06u_tests.log 2>&1 && \
for r in 1 2 3; do timeout -k 10 200 python bench.py --dtype fp8 --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('fp8 bs512', r['ms_per_step'])"; done
echo "exit $?"; tail -3 $O/r06u_tests.log
