#!/usr/bin/env python3
"""Kernels of the main stream inside chosen levels of the last bench step of a rocprofv3 rocpd database.
usage: python tools/level_kernels.py results.db level [level ...]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name,d.start,d.end,d.stream_id,d.grid_size_x,d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
sel = rows[[i for i, r in enumerate(rows) if 'k_init_keys' in r[0]][-1]:]
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", n); return m.group(1) if m else n[:30]
main = [r for r in sel if 'k_split_emit' in r[0]][0][3]
want = set(int(x) for x in sys.argv[2:])
lvl = -1; t0 = None
for r in sel:
    if r[3] == main and 'k_scan_pair' in r[0]:
        lvl += 1; t0 = r[1]
    if lvl in want:
        print("L%d %s %-46s start %8.1f dur %8.1f us  wgs %d" % (lvl, "main" if r[3] == main else "s%-3d" % r[3], short(r[0]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[4] // max(r[5], 1)))
