#!/bin/bash
# GPU call 7: stride-2 data gradient on the XCD-aware one-dimensional grid: parity, per-op times, PMC bytes of the three launches
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dconv_gpu.py -x -q -k "stride2" > $O/r06g_pytest.txt 2>&1; tail -3 $O/r06g_pytest.txt
timeout -k 10 200 python tools/s2_time.py dgrad > $O/r06g_s2_time_dgrad.txt 2>&1; cat $O/r06g_s2_time_dgrad.txt
timeout -k 10 500 bash tools/ab_env.sh MI355_DCONV_S2 0 1 resnet50 3 > $O/r06g_ab_dconv_s2.txt 2>&1; cat $O/r06g_ab_dconv_s2.txt
