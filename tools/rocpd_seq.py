#!/usr/bin/env python3
"""The dispatches of the kernels whose name contains PATTERN, in launch order, with their durations and the gap in front of each
(rocprofv3 rocpd result).  usage: python tools/rocpd_seq.py x_results.db PATTERN [first [count]]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
sym = [t for t in tabs if "kernel_symbol" in t]
print("# table %s columns %s" % (kd, cols), file=sys.stderr)
if "name" in cols:
    q = "select name, start, end from %s order by start" % kd
else:
    st = sym[0]
    q = "select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, st)
rows = list(c.execute(q))
pat = sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 80
prev_end = None
out = []
for n, s, e in rows:
    if pat in n:
        out.append((n.split("(")[0][-28:], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
for n, d, g in out[first:first + count]:
    print("%-28s %9.2f us   gap before %7.2f us" % (n, d, g))
