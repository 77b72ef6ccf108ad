#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --dtype fp8 --batch 512 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary"
rm -rf $O/r06v
MI355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06v -- $S > /dev/null 2> $O/r06v.err
python tools/timeline.py $O/r06v > $O/r06v_timeline_fp8_serial.txt 2>&1; head -75 $O/r06v_timeline_fp8_serial.txt | cut -c1-150
rm -rf $O/r06v
