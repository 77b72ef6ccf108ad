#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
B="python bench.py --dtype fp8 --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for r in 1 2 3; do
  MI355RN_LIB=$PWD/sota_imagenet_amd/lib/libmi355rn_old.so timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('old', r['ms_per_step'])"
  timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('new', r['ms_per_step'])"
done
