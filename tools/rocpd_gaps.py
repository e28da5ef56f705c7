#!/usr/bin/env python3
"""Where the GPU waits: the gaps above MIN us between consecutive kernel dispatches of a rocprofv3 rocpd result, with the kernels on either side
(copies are not in the kernel table: a gap may be a copy).  usage: python tools/rocpd_gaps.py x_results.db [MIN_US [first_kernel_pattern]]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, sym)))
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
pat = sys.argv[3] if len(sys.argv) > 3 else "k_hist256"
def short(n):
    n = n.split("(")[0]
    return n[n.rfind("k_") if "k_" in n else 0:][:30]
starts = [i for i, r in enumerate(rows) if pat in r[0]]
if len(starts) < 2:
    print("pattern not found twice"); sys.exit(0)
a, b = starts[-2], starts[-1]          # the last complete step
step = rows[a:b]
t0 = step[0][1]
busy = 0.0; end = step[0][2]; prev = step[0]
print("step of %d dispatches, %.2f ms from first start to next step's first start" % (len(step), (rows[b][1] - t0) / 1e6))
for r in step[1:] + [rows[b]]:
    g = (r[1] - end) / 1e3
    if g > mn:
        print("  +%8.2f ms  gap %8.1f us  after %-30s before %s" % ((end - t0) / 1e6, g, short(prev[0]), short(r[0])))
    if r[2] > end:
        end = r[2]; prev = r
ksum = sum((r[2] - r[1]) for r in step) / 1e6
print("sum of kernel durations %.2f ms" % ksum)
