#!/usr/bin/env python3
"""main-stream kernel sequence (duration, gap in front) of chosen levels of the last timed bench step in a rocprofv3 database
usage: python tools/level_anatomy.py x_results.db [level ...]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_init_keys" in r[0]]
sel = rows[idx[-3]:idx[-2]] if len(idx) >= 4 else rows[idx[-1]:]
main = sel[0][3]
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", n)
    return m.group(1) if m else n[:30]
ms = [(short(n), s, e) for n, s, e, st in sel if st == main]
other = [(short(n), s, e) for n, s, e, st in sel if st != main]
sc = [i for i, (n, s, e) in enumerate(ms) if n.startswith("k_scan_pair") or n.startswith("k_multi_pick1")]
for L in [int(x) for x in sys.argv[2:]] or [5, 20]:
    if L + 1 >= len(sc):
        continue
    a, b = sc[L], sc[L + 1]
    print("level %d: %.1f us" % (L, (ms[b][1] - ms[a][1]) / 1e3))
    prev = None
    for n, s, e in ms[a:b]:
        gap = (s - prev) / 1e3 if prev else 0.0
        side = [o for o in other if o[1] < s and o[2] > (prev or s)] if gap > 3 else []
        print("   gap %6.1f  %-34s %7.1f   %s" % (gap, n, (e - s) / 1e3, ("(other stream during the gap: " + ", ".join("%s %.0f us" % (o[0], (o[2] - o[1]) / 1e3) for o in side) + ")") if side else ""))
        prev = e
