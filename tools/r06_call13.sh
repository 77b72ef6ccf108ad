#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=15 > $O/r06m_gpu_suite.log 2>&1 && \
timeout -k 10 400 python bench.py > $O/r06m_bench_default.json 2> $O/r06m_bench_default.err && \
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r06m_smoke.log 2>&1
echo "exit $?"; tail -25 $O/r06m_gpu_suite.log; cut -c1-400 $O/r06m_bench_default.json; tail -2 $O/r06m_smoke.log
