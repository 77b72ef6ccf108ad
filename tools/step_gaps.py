"""GPU idle gaps inside the LAST train step of a rocprofv3 --kernel-trace run (steady state needs >= 10 steps: the host must have
filled the launch queue).   python tools/step_gaps.py <trace dir> [min_gap_us]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
sg = [i for i, e in enumerate(ev) if "sgd_kernel" in e[2]]
for which in (-1, len(sg) // 2):
    a, b = sg[which - 1] + 1, sg[which] + 1
    step = ev[a:b]
    end, idle, gaps = step[0][1], 0.0, []
    for s, e, n in step[1:]:
        if s > end:
            idle += (s - end) / 1e3
            if (s - end) / 1e3 >= thr:
                gaps.append((round((s - end) / 1e3, 1), n[:50]))
        end = max(end, e)
    print(f"step {which % len(sg)} of {len(sg)}: wall {(step[-1][1] - step[0][0]) / 1e6:.3f} ms, kernels {len(step)}, idle {idle / 1e3:.3f} ms, gaps >= {thr} us: {gaps}")
