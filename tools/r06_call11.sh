#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py -x -q -k "side_stream or stem_backward or teacher_forced_layers or gradient_accumulation or fp32_forward_backward or bf16_forward_backward" > $O/r06k_pytest.txt 2>&1; tail -4 $O/r06k_pytest.txt
for r in 1 2 3; do timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bf16 bs256', r['ms_per_step'])"; done 2>&1 | tee $O/r06k_bench.txt
rm -rf $O/r06k_tr; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06k_tr -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2> $O/r06k_tr.err
python tools/queue_busy.py $O/r06k_tr | tee $O/r06k_queue_busy.txt; rm -rf $O/r06k_tr
