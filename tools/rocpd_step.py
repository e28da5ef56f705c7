#!/usr/bin/env python3
"""The kernels of the LAST step of a profiled bench run in launch order (rocprofv3 rocpd result, view `kernels`): consecutive launches of the same
kernel are one line (count, total, longest); launches on other streams are merged by start time.
usage: python tools/rocpd_step.py x_results.db [marker-kernel, default k_hist256] [min_us]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:48]


c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_hist256"
rows = [(short(n), s, e) for n, s, e in c.execute("select name,start,end from kernels order by start")]
idx = [i for i, r in enumerate(rows) if marker in r[0]]
sel = rows[idx[-1]:] if idx else rows
t0 = sel[0][1]
out = []
for n, s, e in sel:
    d = (e - s) / 1e3
    if out and out[-1][0] == n:
        out[-1][1] += 1; out[-1][2] += d; out[-1][3] = max(out[-1][3], d); out[-1][5] = (e - t0) / 1e3
    else:
        out.append([n, 1, d, d, (s - t0) / 1e3, (e - t0) / 1e3])
print("# last step: %d launches, %.2f ms from first start to last end, %.2f ms of kernel time" % (len(sel), (sel[-1][2] - t0) / 1e6, sum(e - s for _, s, e in sel) / 1e6))
for n, k, tot, mx, a, b in out:
    print("%9.1f us  %-48s x%-4d total %9.1f us  longest %8.1f" % (a, n, k, tot, mx))
