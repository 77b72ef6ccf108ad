#!/bin/bash
# GPU call 2 of round 6: clean MFMA-shape microbenchmark; parity of the bn-in-operand-path kernels; executor tests; in-step A/B of MI355_DCONV_BN
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_shape_random.hip -o /tmp/mfma_shape_random 2>/dev/null && timeout -k 10 120 /tmp/mfma_shape_random > $O/r06_mfma_shape_random_data.txt 2>&1
cat $O/r06_mfma_shape_random_data.txt
timeout -k 10 500 python -m pytest tests/test_dconv_gpu.py -x -q -k "bn1 or stride2" > $O/r06b_pytest_bnin.txt 2>&1; tail -5 $O/r06b_pytest_bnin.txt
timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py -x -q -k "baseline_batch or segment_by_segment or teacher_forced_layers" > $O/r06b_pytest_exec.txt 2>&1; tail -8 $O/r06b_pytest_exec.txt
timeout -k 10 500 bash tools/ab_env.sh MI355_DCONV_BN 0 1 resnet50 3 > $O/r06b_ab_dconv_bn.txt 2>&1; cat $O/r06b_ab_dconv_bn.txt
