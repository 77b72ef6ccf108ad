#!/bin/bash
# GPU call 1 of round 6: MFMA shape on random data, parity of the stride-2 data-gradient kernels, per-op and in-step A/B
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_shape_random.hip -o /tmp/mfma_shape_random && timeout -k 10 120 /tmp/mfma_shape_random > $O/r06_mfma_shape_random_data.txt 2>&1
cat $O/r06_mfma_shape_random_data.txt
timeout -k 10 500 python -m pytest tests/test_dconv_gpu.py -x -q -k "stride2" > $O/r06a_pytest_s2d.txt 2>&1; tail -5 $O/r06a_pytest_s2d.txt
timeout -k 10 400 python -m pytest tests/test_resnet_gpu.py -x -q -k "baseline_batch or segment_by_segment" > $O/r06a_pytest_exec.txt 2>&1; tail -5 $O/r06a_pytest_exec.txt
timeout -k 10 200 python tools/s2_time.py dgrad > $O/r06a_s2_time_dgrad.txt 2>&1; cat $O/r06a_s2_time_dgrad.txt
timeout -k 10 400 bash tools/ab_env.sh MI355_DCONV_S2 0 1 resnet50 3 > $O/r06a_ab_dconv_s2_dgrad.txt 2>&1; cat $O/r06a_ab_dconv_s2_dgrad.txt
