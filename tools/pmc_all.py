#!/usr/bin/env python3
"""HBM bytes per launch of EVERY kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output), next to the
kernel's average duration from the same passes' kernel trace.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIRF -o f -- python bench.py --steps 2 --warmup 1 --no-cpu --no-check
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIRW -o w -- python bench.py --steps 2 --warmup 1 --no-cpu --no-check
  python tools/pmc_all.py DIRF DIRW [rows] > profiles/rNN_pmc_traffic_c4.txt

Corrections as MI355X_MICROARCH.md's HBM section prescribes (and tools/pmc_summary.py applies for the judged kernel): the
counters are in KB (x1024); FETCH_SIZE x2 on gfx950; WRITE_SIZE as is.  Durations under counter collection are longer than in a
plain trace (the dispatches are serialised); GB/s here = corrected bytes / that duration, so it is a lower bound."""
import csv
import glob
import os
import re
import sys


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", k)[:48]


def collect(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        disp = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                key = (f, r["Dispatch_Id"])
                e = disp.setdefault(key, [short(r["Kernel_Name"]), 0.0])
                e[1] += float(r["Counter_Value"])
        for name, v in disp.values():
            per.setdefault(name, []).append(v)
    return per


def durations(d):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def main():
    df, dw = sys.argv[1], sys.argv[2]
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    fe, wr, du = collect(df, "FETCH_SIZE"), collect(dw, "WRITE_SIZE"), durations(df)
    out = []
    for k in set(fe) | set(wr):
        f = fe.get(k, [0.0]); w = wr.get(k, [0.0]); t = du.get(k, [0.0])
        fb = sum(f) / len(f) * 1024 * 2
        wb = sum(w) / len(w) * 1024
        us = sum(t) / len(t) if t else 0.0
        out.append(((fb + wb) * len(f), k, len(f), fb, wb, us))
    print("# FETCH_SIZE (x1024 x2) / WRITE_SIZE (x1024) per launch, separate passes; us = average duration in the FETCH pass")
    print("%-50s %6s %12s %12s %12s %10s %9s" % ("kernel", "calls", "fetch_MB", "write_MB", "total_MB", "avg_us", "GB/s"))
    for _, k, n, fb, wb, us in sorted(out, reverse=True)[:rows]:
        print("%-50s %6d %12.2f %12.2f %12.2f %10.1f %9.0f" % (k, n, fb / 1e6, wb / 1e6, (fb + wb) / 1e6, us, (fb + wb) / 1e3 / us if us else 0))


if __name__ == "__main__":
    main()
