#!/usr/bin/env python3
"""Latency of one small alignment (reveal/refine.py feeds rem.align bubbles of a few hundred bases; SURVEY 3.2 calls these
calls latency-bound): construct + recursion for two sequences of L bases, 1 % substitutions --
  builtin   index.construct() + index.align_builtin(20, 2)          (the library alone)
  rem.align reveal_amd.rem.align([(name, seq), ...], minlength=20)  (graph callbacks in Python on top)
  whole     a fresh index per alignment, from reveallib.index() to its release (what a loop over bubbles pays per bubble: the handle's
            streams and pinned buffers come from a process-wide pool -- created and destroyed per handle they cost 6 ms)
median of `reps` runs after a warm-up run, milliseconds.  usage (GPU box): python tools/latency_probe.py [reps]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reveal_amd import rem, reveallib, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print("%8s %14s %14s %14s %14s %10s" % ("L", "construct_ms", "builtin_ms", "whole_ms", "rem.align_ms", "anchors"))
for L in (100, 300, 1000, 3000, 10000, 100000):
    seqs = [s.decode() for s in synth.genomes(L, 2, seed=9)]
    tc, tb, tr, tw = [], [], [], []
    na = 0
    for rep in range(reps + 1):
        tn = time.perf_counter()
        idx = reveallib.index()
        for k, s in enumerate(seqs):
            idx.addsample("s%d" % k)
            idx.addsequence(s)
        t0 = time.perf_counter()
        idx.construct()
        t1 = time.perf_counter()
        res = idx.align_builtin(20, 2)
        t2 = time.perf_counter()
        na = len(res["anchors"][0])
        del idx, res
        tf = time.perf_counter()
        G, _ = rem.align([("s0", seqs[0]), ("s1", seqs[1])], minlength=20)
        t3 = time.perf_counter()
        if rep:
            tc.append((t1 - t0) * 1e3); tb.append((t2 - t0) * 1e3); tw.append((tf - tn) * 1e3); tr.append((t3 - tf) * 1e3)
    print("%8d %14.3f %14.3f %14.3f %14.3f %10d" % (L, statistics.median(tc), statistics.median(tb), statistics.median(tw), statistics.median(tr), na))
