"""Per-kernel SQ counter summary from a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run (north_star: rocprof-reported
MFMA utilisation of the conv kernels).  usage: python tools/sq_counters.py <dir> [name filter ...]
  MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed
  over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'); the SQ wait / active counters are reported as a share of SQ_WAVE_CYCLES
  (quad-cycle units cancel)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
filt = sys.argv[2:] or ["dconv_", "pw_k", "igemm8_kernel", "igemm_kernel", "wgrad_kernel"]
f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter -> value
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not any(x in k for x in filt):
        continue
    per[k][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
CUS = 256
print(f"{'kernel':58s} {'n':>4s} {'us':>7s} {'clkGHz':>6s} {'MFMAbusy%':>9s} {'waitAny%':>8s} {'waitInst%':>9s} {'waitLDS%':>8s} {'actLDS%':>7s}")
for k, disp in sorted(per.items()):
    rows = [v for v in disp.values() if "GRBM_GUI_ACTIVE" in v]
    if not rows:
        continue
    rows = rows[len(rows) // 4:]  # skip warm-up launches
    avg = lambda c: sum(v.get(c, 0.0) for v in rows) / len(rows)
    cyc = avg("GRBM_GUI_ACTIVE") / 8
    wave = max(avg("SQ_WAVE_CYCLES"), 1.0)
    name = k if len(k) <= 58 else k[:55] + "..."
    print(f"{name:58s} {len(rows):4d} {'':>7s} {'':>6s} {100 * avg('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * CUS * cyc):9.1f} "
          f"{100 * avg('SQ_WAIT_ANY') / wave:8.1f} {100 * avg('SQ_WAIT_INST_ANY') / wave:9.1f} {100 * avg('SQ_WAIT_INST_LDS') / wave:8.1f} "
          f"{100 * avg('SQ_ACTIVE_INST_LDS') / wave:7.1f}   cycles/launch {cyc:9.0f}")
