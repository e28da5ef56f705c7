#!/usr/bin/env python3
"""Same-run A/B of one switch (rv_set_option): construct + align_builtin of a synthetic workload, alternating the values step by step.
    python tools/ab_opt.py --L 250000000 --base RV_NO_CASCADE=1 --opt RV_SPLIT_XCD=0,1 [--genomes 2 --steps 3 --indelfrac 0]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reveal_amd import reveallib, reveallib64, synth

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=250_000_000); ap.add_argument("--genomes", type=int, default=2); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--indelfrac", type=float, default=0.0); ap.add_argument("--sa64", action="store_true")
ap.add_argument("--base", default=""); ap.add_argument("--opt", required=True)
a = ap.parse_args()
seqs = synth.genomes(a.L, a.genomes, seed=42, indelfrac=a.indelfrac)
idx = (reveallib64 if a.sa64 else reveallib).index()
for k, s in enumerate(seqs):
    idx.addsample("g%d" % k); idx.addsequence(s)
for kv in [x for x in a.base.split(",") if x]:
    k, v = kv.split("="); idx.set_option(k, int(v))
name, vals = a.opt.split("=")
vals = [int(v) for v in vals.split(",")]
res = {v: [] for v in vals}
prof = {}
idx.construct(); idx.align_builtin(20, 2)
for it in range(a.steps):
    for v in vals:
        idx.set_option(name, v)
        idx.prof(enable=True, reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.construct(); r = idx.align_builtin(20, 2)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) * 1e3)
        prof[v] = {k: round(x[1], 2) for k, x in idx.prof(enable=False).items() if x[0]}
for v in vals:
    print("%s=%d  ms/step %s  (min %.2f)  classes %s" % (name, v, " ".join("%.2f" % x for x in res[v]), min(res[v]), prof[v]), flush=True)
