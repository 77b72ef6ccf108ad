#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for r in 1 2; do
for cfg in "MI355_BN_FUSE_FIN=0" "MI355_BN_FUSE_FIN=1" "MI355_BN_FUSE_FIN=1 MI355_FIN_NOWAIT=1"; do
  env $cfg timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$cfg', r['ms_per_step'])"
done; done
