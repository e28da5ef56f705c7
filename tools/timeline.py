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




def gaps(path, top=12):
    """main-stream view of the last step: kernel time, and idle time attributed to the kernel that ran before each gap"""
    c = sqlite3.connect(path)
    rows = list(c.execute("select name,start,end,stream_id from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "k_init_keys" in r[0]]
    # bench.py ends with two untimed steps that time every kernel class: look at the last step of the timed region
    sel = rows[idx[-3]:idx[-2]] if len(idx) >= 4 else rows[idx[-1]:]
    # the main stream = the one k_init_keys ran on
    main = sel[0][3]
    ms = [(short(n), s, e) for n, s, e, st in sel if st == main]
    busy = sum(e - s for _, s, e in ms)
    span = ms[-1][2] - ms[0][1]
    by = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(ms, ms[1:]):
        g = max(0, s1 - e0)
        k = n0 + " -> " + n1
        by[k] = by.get(k, 0) + g
    print("main stream: %d kernels, span %.3f ms, kernels %.3f ms, gaps %.3f ms" % (len(ms), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:top]:
        print("  %-60s %8.1f us" % (k, v / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
        gaps(sys.argv[1])
    else:
        main()
