#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) into profiles/pmc_scan.json.

Collected exactly as MI355X_MICROARCH.md's HBM section prescribes -- one counter per pass, no trace domains
besides --kernel-trace:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write k_scan_pair "<workload>" > profiles/pmc_scan.json

Corrections (same guide): the counters are in KB (x1024); on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced streaming read (x2); WRITE_SIZE is used as is.
"""
import csv
import glob
import json
import os
import sys


def collect(d, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            hit = (kernel in kn) if "<" in kernel else ((kernel + "(") in kn.replace("<", "("))      # "k_full_scan<0>": that instance only
            if r["Counter_Name"] == counter and hit:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals += list(per.values())
    return vals


def main():
    dfetch, dwrite, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
    workload = sys.argv[4] if len(sys.argv) > 4 else ""
    fe = collect(dfetch, "FETCH_SIZE", kernel)
    wr = collect(dwrite, "WRITE_SIZE", kernel)
    if not fe or not wr:
        sys.exit("no %s dispatches found" % kernel)
    fetch_kb = sum(fe) / len(fe)
    write_kb = sum(wr) / len(wr)
    out = {
        "kernel": kernel,
        "workload": workload,
        "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), see tools/pmc_summary.py",
        "correction": "KB -> x1024; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is",
        "launches": len(fe),
        "avg_FETCH_SIZE_KB": fetch_kb,
        "avg_WRITE_SIZE_KB": write_kb,
        "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
