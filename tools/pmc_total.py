#!/usr/bin/env python3
"""Whole-step HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes) of ANY bench.py command, every kernel counted:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024          (MI355X_MICROARCH.md, HBM / rocprofv3 section: KiB units, the gfx950 half-count of reads)
per kernel symbol and per step (steps = launches of the once-per-step SGD kernel).  tools/pmc_traffic.py is the per-class form for the ResNet-50 step;
this one serves the BResNet-50 step (configs[3]), whose kernels that tool does not name.
    python tools/pmc_total.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> > profiles/<round>_pmc_traffic_bresnet50_bf16.json"""
import collections
import csv
import glob
import json
import os
import re
import sys


def name_of(r):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("mi355::", "")
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n).strip()[:100]   # (drop the parameter list)


def load(d, counter):
    """the dispatches of the LAST whole step (from behind the second-to-last SGD launch up to the last one): [(kernel, counter value)]"""
    f = glob.glob(f"{d}/pmc_{counter}/**/*_counter_collection.csv", recursive=True)[0]
    rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
    assert len(sgd) >= 2, "fewer than two SGD launches in the pass"
    return [(name_of(r), float(r["Counter_Value"])) for r in rows[sgd[-2] + 1: sgd[-1] + 1]]


def main():
    d = sys.argv[1]
    fe, wr = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    assert [k for k, _ in fe] == [k for k, _ in wr], "the two passes dispatched different kernel sequences"
    ker = collections.defaultdict(lambda: [0, 0.0])
    for (k, f), (_, w) in zip(fe, wr):
        ker[k][0] += 1
        ker[k][1] += (2 * f + w) * 1024
    total = int(sum(v[1] for v in ker.values()))
    top = {k: {"launches": v[0], "hbm_bytes": int(v[1])} for k, v in sorted(ker.items(), key=lambda kv: -kv[1][1])[:40]}
    print(json.dumps({"formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 half-count correction), every kernel of the last whole step of the passes",
                      "commit": os.environ.get("MI355_PROFILE_COMMIT"), "kernels_in_step": len(fe), "hbm_bytes_per_step": total, "kernels_top40": top}, indent=1))


if __name__ == "__main__":
    main()
