#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for cfg in "fp8 512" "bf16 512" "bf16 256"; do set -- $cfg
  rm -rf $O/r06j_tr
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06j_tr -- python3 bench.py --dtype $1 --batch $2 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2> $O/r06j_tr.err
  echo "== $1 bs $2"; python tools/queue_busy.py $O/r06j_tr
  rm -rf $O/r06j_tr
done 2>&1 | tee $O/r06j_queue_busy.txt
