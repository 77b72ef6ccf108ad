#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for r in 1 2 3; do for v in 0 2 3; do
  MI355_DCONV_BN=$v timeout -k 10 200 $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('DCONV_BN=$v', r['ms_per_step'])"
done; done
