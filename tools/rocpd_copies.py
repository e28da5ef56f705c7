#!/usr/bin/env python3
"""Memory copies above MIN us in a rocprofv3 rocpd result (--memory-copy-trace), with the kernels that end/start around them.
usage: python tools/rocpd_copies.py x_results.db [MIN_US]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
mc = [t for t in tabs if "memory_cop" in t]
print("# tables", mc, file=sys.stderr)
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
t = mc[0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
print("# columns", cols, file=sys.stderr)
rows = list(c.execute("select * from %s order by start" % t))
si, ei = cols.index("start"), cols.index("end")
szi = cols.index("size") if "size" in cols else None
t0 = rows[0][si] if rows else 0
for r in rows:
    d = (r[ei] - r[si]) / 1e3
    if d >= mn and (len(sys.argv) < 4 or r[szi] != int(sys.argv[3])):
        print("copy at +%10.2f ms  %9.1f us  %s bytes" % ((r[si] - t0) / 1e6, d, r[szi] if szi is not None else "?"))
