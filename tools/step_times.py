#!/usr/bin/env python3
"""per-step wall times of construct + align_builtin on a synthetic workload: python tools/step_times.py --L 5000000 --genomes 10 --steps 20"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reveal_amd import reveallib, synth
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=5_000_000); ap.add_argument("--genomes", type=int, default=10); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
seqs = synth.genomes(a.L, a.genomes, seed=42)
idx = reveallib.index()
for k, s in enumerate(seqs):
    idx.addsample("g%d" % k); idx.addsequence(s)
ts = []
for it in range(a.steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.construct(); t1 = time.perf_counter()
    r = idx.align_builtin(20, 2)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print(" ".join("%.1f+%.1f" % t for t in ts))
