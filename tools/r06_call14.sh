#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py -x -q -k "finalize_inside or side_stream_is_bitwise or loss_curve_matches_cpu" > $O/r06n_tests.log 2>&1 && \
bash tools/ab_env.sh MI355_BN_FUSE_FIN 0 1 resnet50 3 > $O/r06n_ab_fuse_fin.txt 2>&1
echo "exit $?"; tail -8 $O/r06n_tests.log; cat $O/r06n_ab_fuse_fin.txt
