#!/usr/bin/env python3
"""Per-launch view of the last bench step in a rocprofv3 rocpd database (--kernel-trace):
durations of selected kernels in launch order, and how long no kernel was running on any stream.
usage: python tools/timeline.py gpurun_out/profNN/*/*_results.db [kernel ...]"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"(k_\w+|__amd\w+)", n)
    return m.group(1) if m else n[:30]


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select name,start,end,grid_x,scratch_size,vgpr_count,lds_size from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "k_init_keys" in r[0]]
    sel = rows[idx[-1]:]
    names = sys.argv[2:] or ["k_tile_carry", "k_split_count", "k_split_emit", "k_scan_pair", "k_bubble_child", "k_leaf"]
    for name in names:
        d = [round((e - s) / 1e3, 1) for (n, s, e, g, sc, vg, lds) in sel if short(n) == name]
        meta = [(sc, vg, lds) for (n, s, e, g, sc, vg, lds) in sel if short(n) == name][:1]
        print(name, "scratch/vgpr/lds", meta, d)
    iv = sorted((s, e) for n, s, e, *_ in sel)
    idle, ce = 0, iv[0][0]
    for s, e in iv:
        if s > ce:
            idle += s - ce
        ce = max(ce, e)
    print("%d kernels in the last step, span %.3f ms, no kernel running for %.3f ms" % (len(sel), (ce - iv[0][0]) / 1e6, idle / 1e6))


if __name__ == "__main__":
    main()
