import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t]
ks = [t for t in tabs if t.startswith('kernels') or t == 'kernels']
print(kd[:3], ks[:3])
try:
    rows = list(c.execute("select name, start, end from kernels order by start"))
except Exception as e:
    print("ERR", e); print(tabs); sys.exit()
sel = [(n, s, e) for n, s, e in rows if sys.argv[2] in n]
d = [(e - s) / 1e3 for n, s, e in sel]
print(len(d)); print(" ".join("%.0f" % x for x in d[-38:]))
