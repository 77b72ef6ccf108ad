#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dconv_gpu.py tests/test_variant_gpu.py -x -q -k "leaky or bn1_bn2 or conv1_data_gradient or oracle" > $O/r06q_tests.log 2>&1
echo "exit $?"; tail -12 $O/r06q_tests.log
