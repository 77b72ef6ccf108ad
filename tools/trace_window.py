"""print the kernels of the last step of a rocprofv3 kernel trace between the n-th and m-th launch: start offset, duration, gap to the previous end, stream, name"""
import csv, glob, sys
d, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_ingest" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]["Start_Timestamp"]); prev = t0
for i, r in enumerate(step[a:b], a):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%4d  +%9.1f us  dur %7.1f  gap %6.1f  q%s  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Queue_Id"], r["Kernel_Name"][:70]))
    prev = e
