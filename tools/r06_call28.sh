#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --model bresnet50 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary"
rm -rf $O/r06z
MI355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06z -- $S > /dev/null 2> $O/r06z.err
f=$(ls $O/r06z/*/*_kernel_stats.csv | head -1); cp $f $O/r06z_bresnet50_kernel_stats_serial_final.csv
python tools/timeline.py $O/r06z > $O/r06z_bresnet50_timeline_serial_final.txt; head -14 $O/r06z_bresnet50_timeline_serial_final.txt | cut -c1-120
rm -rf $O/r06z
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06z -- $S > $O/r06z_bench.jsonl 2> $O/r06z.err
python tools/timeline.py $O/r06z > $O/r06z_bresnet50_timeline_final.txt; head -3 $O/r06z_bresnet50_timeline_final.txt | cut -c1-120
python tools/queue_busy.py $O/r06z > $O/r06z_bresnet50_queue_busy_final.txt; cat $O/r06z_bresnet50_queue_busy_final.txt
rm -rf $O/r06z
