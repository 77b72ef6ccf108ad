#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
MI355_TRACE_KERNELS=1 timeout -k 10 300 python tools/bres_fuse_check.py 16 > $O/r06o_fuse_check.txt 2> $O/r06o_trace.txt
echo "exit $?"; tail -3 $O/r06o_fuse_check.txt; grep -c "_s3" $O/r06o_trace.txt; grep "_s3" $O/r06o_trace.txt | sort | uniq -c | sort -rn | head -20
timeout -k 10 600 python -m pytest tests/test_variant_gpu.py -x -q > $O/r06o_variant_tests.log 2>&1; tail -3 $O/r06o_variant_tests.log
bash tools/ab_env.sh MI355_BRESNET_FUSE_BN_BWD 0 1 bresnet50 3 > $O/r06o_ab.txt 2>&1; cat $O/r06o_ab.txt
