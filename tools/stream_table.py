"""stream_table.py <serial trace dir> <overlapped trace dir> [kernels per step] — per kernel symbol: launches of the last step, summed duration with
the weight-gradient side stream off (MI355_WGRAD_STREAM=0) and in the default two-stream step, and the ratio: which kernels pay for sharing the
chip with the other stream (rocprofv3 --kernel-trace csv, as tools/profile_round.sh writes them)."""
import csv
import glob
import re
import sys


def last_step(d, n):
    f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows[-n:]


def short(name):
    name = re.sub(r"^_ZN5mi35512_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"void mi355::\(anonymous namespace\)::", "", name)
    return name[:44]


def table(rows):
    t = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        a = t.setdefault(k, [0, 0.0, 0.0])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = max(a[2], d)
    return t


if __name__ == "__main__":
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 445
    ser, ovl = table(last_step(sys.argv[1], n)), table(last_step(sys.argv[2], n))
    print("%-46s %5s %10s %10s %6s %9s" % ("kernel", "n", "serial us", "in-step us", "ratio", "max us"))
    tot_s = tot_o = 0.0
    for k in sorted(ovl, key=lambda k: -(ovl[k][1] - ser.get(k, [0, 0, 0])[1])):
        s = ser.get(k, [0, 0.0, 0.0])
        tot_s += s[1]
        tot_o += ovl[k][1]
        print("%-46s %5d %10.1f %10.1f %6.2f %9.1f" % (k, ovl[k][0], s[1], ovl[k][1], ovl[k][1] / s[1] if s[1] else 0.0, ovl[k][2]))
    print("%-46s %5s %10.1f %10.1f" % ("sum of kernel durations", "", tot_s, tot_o))
