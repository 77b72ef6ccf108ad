#!/bin/bash
# GPU call 8: generated e4m3 3x3 kernels: parity, the fp8 step tests, fp8 vs bf16 at bs 512 with MI355_DCONV_FP8=0 / 1
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_fp8_gpu.py -x -q > $O/r06h_pytest_fp8.txt 2>&1; tail -4 $O/r06h_pytest_fp8.txt
timeout -k 10 600 python -m pytest tests/test_fp8_step_gpu.py -x -q > $O/r06h_pytest_fp8_step.txt 2>&1; tail -4 $O/r06h_pytest_fp8_step.txt
for r in 1 2; do for v in 0 1; do
  MI355_DCONV_FP8=$v timeout -k 10 300 python bench.py --dtype fp8 --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('fp8 bs512 MI355_DCONV_FP8=$v', r['ms_per_step'])"
done; done 2>&1 | tee $O/r06h_ab_dconv_fp8.txt
timeout -k 10 300 python bench.py --dtype bf16 --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bf16 bs512', r['ms_per_step'])" | tee -a $O/r06h_ab_dconv_fp8.txt
