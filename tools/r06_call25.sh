#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_variant_gpu.py -x -q -s -k "teacher_forced" > $O/r06w_tests.log 2>&1
echo "exit $?"; tail -25 $O/r06w_tests.log
