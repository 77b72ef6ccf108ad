#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --model bresnet50 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary"
rm -rf $O/r06s
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r06s -- $S > /dev/null 2> $O/r06s.err
python tools/queue_busy.py $O/r06s > $O/r06s_queue_busy_bresnet.txt 2>&1; cat $O/r06s_queue_busy_bresnet.txt
python tools/timeline.py $O/r06s > $O/r06s_timeline_bresnet.txt 2>&1; head -30 $O/r06s_timeline_bresnet.txt | cut -c1-130
f=$(ls $O/r06s/*/*_kernel_trace.csv | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sgd=[i for i,r in enumerate(rows) if "sgd" in r["Kernel_Name"]]
a,b=sgd[-2]+1,sgd[-1]+1
step=rows[a:b]; t0=int(step[0]["Start_Timestamp"])
qs=sorted({r["Queue_Id"] for r in step})
# gaps on the main queue (the one with most kernels) longer than 15 us, with the kernels around them
from collections import Counter
mq=Counter(r["Queue_Id"] for r in step).most_common(1)[0][0]
m=[r for r in step if r["Queue_Id"]==mq]
for x,y in zip(m,m[1:]):
    gap=(int(y["Start_Timestamp"])-int(x["End_Timestamp"]))/1e3
    if gap>15: print("gap %6.1f us at %7.3f ms  after %-50s before %-50s"%(gap,(int(x["End_Timestamp"])-t0)/1e6,x["Kernel_Name"][:50],y["Kernel_Name"][:50]))
PY
rm -rf $O/r06s
