#!/bin/bash
# On the GPU box: same-box A/B of one environment switch on the default bench line (interleaved A B A B): tools/ab_env.sh VAR A_VALUE B_VALUE [bench args]
VAR=${1:?var}; A=${2:?a}; B=${3:?b}; shift 3
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    env $VAR=$v python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$VAR=$v', d['ms_per_step'])"
  done
done
