#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite result (--kernel-trace --stats) as a text table.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/rNN_*.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:64]


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats   (%s)" % path)
    print("# durations in microseconds; total GPU kernel time %.3f ms" % (tot / 1e3))
    print("%-64s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for n, calls, total, avg, pct in rows:
        print("%-64s %8d %14.1f %12.2f %7.2f" % (short(n), calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1])
