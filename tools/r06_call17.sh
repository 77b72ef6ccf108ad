#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
S="python3 bench.py --model bresnet50 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary"
for v in 0 1; do
rm -rf $O/r06p_$v
MI355_WGRAD_STREAM=0 MI355_BRESNET_FUSE_BN_BWD=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06p_$v -- $S > /dev/null 2> $O/r06p_$v.err
f=$(ls $O/r06p_$v/*/*_kernel_stats.csv | head -1); cp $f $O/r06p_bres_fuse${v}_kernel_stats.csv; rm -rf $O/r06p_$v
done
python - <<'PY'
import csv
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        d[r['Name']]=(int(r['Calls']), int(r['TotalDurationNs']))
    return d
a=load('gpurun_out/r06p_bres_fuse0_kernel_stats.csv'); b=load('gpurun_out/r06p_bres_fuse1_kernel_stats.csv')
names=sorted(set(a)|set(b), key=lambda n: -abs(a.get(n,(0,0))[1]-b.get(n,(0,0))[1]))
ta=sum(v[1] for v in a.values()); tb=sum(v[1] for v in b.values())
print("total ms per step: unfused %.3f fused %.3f (11 steps)"%(ta/11/1e6, tb/11/1e6))
for n in names[:30]:
    ca,da=a.get(n,(0,0)); cb,db=b.get(n,(0,0))
    print("%-60s  %4d %9.1f us | %4d %9.1f us | diff/step %8.1f us"%(n[:60], ca, da/max(ca,1)/1e3, cb, db/max(cb,1)/1e3, (db-da)/11/1e3))
PY
