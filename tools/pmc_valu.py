#!/usr/bin/env python3
"""VALU instructions per kernel from a rocprofv3 PMC pass (csv):
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES VALUBusy --output-format csv -d DIR -o p -- python bench.py --steps 2 --warmup 1 --no-cpu
  python tools/pmc_valu.py DIR"""
import csv, glob, re, sys
agg = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)[:44]
        agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
rows = []
for k, v in agg.items():
    iv = v.get("SQ_INSTS_VALU", [0]); w = v.get("SQ_WAVES", [0]); vb = v.get("VALUBusy", [0])
    rows.append((sum(iv), k, len(iv), sum(iv) / len(iv), sum(w) / len(w), sum(vb) / len(vb)))
print("%-46s %6s %14s %10s %10s %9s" % ("kernel", "calls", "VALU/launch", "waves", "VALU/wave", "VALUBusy%"))
for t, k, n, a, w, vb in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-46s %6d %14.0f %10.0f %10.0f %9.1f" % (k, n, a, w, a / max(w, 1), vb))
