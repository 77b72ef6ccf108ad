#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_variant_gpu.py -x -q > $O/r06r_tests.log 2>&1
echo "exit $?"; tail -5 $O/r06r_tests.log
bash tools/ab_env.sh MI355_BRESNET_BATCH_PREP 0 1 bresnet50 3 > $O/r06r_ab.txt 2>&1; cat $O/r06r_ab.txt
