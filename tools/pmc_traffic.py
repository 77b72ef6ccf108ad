#!/usr/bin/env python3
"""Condenses two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE — separate passes, TCC slots do not fit both) of
`bench.py` into per-kernel HBM traffic per launch, as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE/WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports exactly half of a 16-byte-per-lane coalesced read
stream, which is what every kernel here issues — checked below on bn_reduce<.,0>, whose algorithmic read is exact).

    cd /tmp && export TMPDIR=/tmp && cd $REPO
    for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- \
        python3 bench.py --steps 2 --warmup 1 --dtype bf16 --no-cpu-baseline --no-roofline --no-secondary; done   (MI355_WGRAD_STREAM=0)
    python tools/pmc_traffic.py gpurun_out bf16 > profiles/<round>_pmc_traffic_bf16.json
"""
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    m0 = re.match(r"(dconv_l\d|pw_k\d+_n\d+|pk_k\d+_n\d+_w\d+|po_k\d+_b\d+|wg3_l\d|wg1_c\d+_o\d+)", n)  # generated assembly kernels (asm/dconv_gen.py, asm/pw_gen.py): one symbol per layer shape
    if m0:
        return m0.group(1)
    m = re.search(r"(igemm8_kernel|igemm_kernel|wgrad_kernel|bn_reduce_kernel|bn_apply_kernel|bn_bwd_apply_kernel|bn_finalize_kernel|"
                  r"splitk_reduce_kernel|sgd_kernel|fc_kernel|dbias_kernel|ce_row_kernel|bn_relu_maxpool\d?_kernel|maxpool_\w+_kernel|gap_\w+_kernel|stem_ingest_kernel|"
                  r"stem_direct_kernel|stem_bwd_reduce_kernel|stem_bwd_apply_kernel|weight_prep_batch_kernel|weight_prep_kernel)", n)
    if not m:
        return None
    base = m.group(1)
    # keep the element type and the tile shape; drop the epilogue / mask variant parameters (one kernel source each)
    keep = {"igemm_kernel": 3, "wgrad_kernel": 3, "igemm8_kernel": 2}.get(base, 1)
    if "bool _Accum" in n:  # rocprofv3's demangler garbles <__bf16, ...>
        rest = re.search(base + r"<bool _Accum, (.*)>", n)
        nums = re.findall(r"\d+", rest.group(1)) if rest else []
        return f"{base}<{','.join(['__bf16'] + nums[:keep - 1])}>" if base != "igemm8_kernel" else f"{base}<{','.join(nums[:keep])}>"
    if "<" in n:  # demangled
        t = re.search(base + r"<([^>]*)>", n)
        args = [a.strip() for a in t.group(1).split(",")] if t else []
        return f"{base}<{','.join(args[:keep])}>" if args else base
    t = re.search(base + r"I(.*?)EEv", n)  # Itanium-mangled template arguments
    if not t:
        return base
    args, rest = [], t.group(1) + "E"
    while rest:
        if rest.startswith("DF16b"):
            args.append("__bf16"); rest = rest[5:]
        elif rest.startswith("f"):
            args.append("float"); rest = rest[1:]
        elif rest.startswith("Li"):
            m2 = re.match(r"Li(\d+)E", rest)
            args.append(m2.group(1)); rest = rest[m2.end():]
        elif rest.startswith("Lb"):
            m2 = re.match(r"Lb(\d)E", rest)
            args.append(m2.group(1)); rest = rest[m2.end():]
        else:
            rest = rest[1:]
    return f"{base}<{','.join(args[:keep])}>"


def load(root, counter):
    f = glob.glob(f"{root}/pmc_{counter}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    root, dtype = sys.argv[1], sys.argv[2]
    fe, wr = load(root, "FETCH_SIZE"), load(root, "WRITE_SIZE")
    out = {"dtype": dtype, "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 half-count correction)",
           "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        n = fe[k][0]
        f_kib = fe[k][1] / n
        w_kib = wr[k][1] / max(wr[k][0], 1) if k in wr else 0.0
        out["kernels"][k] = {"launches": n, "fetch_kib_raw_per_launch": round(f_kib, 1), "write_kib_per_launch": round(w_kib, 1),
                             "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024)}
    # note: with the weight-gradient side stream two kernels can be resident at once; the counters are still attributed
    # per dispatch, so run the PMC passes with MI355_WGRAD_STREAM=0 (serial) for clean per-kernel numbers
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
