#!/bin/bash
# On the GPU box: same-box comparison of several values of one environment switch on the default bench line (interleaved rounds):
#   tools/ab_env3.sh VAR "v0 v1 v2 ..." [bench args]
VAR=${1:?var}; VALS=${2:?values}; shift 2
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
for rep in 1 2 3; do
  for v in $VALS; do
    env $VAR=$v python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$VAR=$v', d['ms_per_step'])"
  done
done
