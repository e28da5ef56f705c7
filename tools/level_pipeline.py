#!/usr/bin/env python3
"""Time construct + the level pipeline (RV_NO_CASCADE=1: every level's scan / split / bubble_sort, leaf kernel on) on synthetic genomes, with the per-class kernel times:
python tools/level_pipeline.py [--L 250000000] [--genomes 2]          (RV_LIB_DIR=<dir> picks another build of the library, e.g. one made with EXTRA=-DRV_LEAF_N=4096)"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reveal_amd import reveallib, synth
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=250_000_000); ap.add_argument("--genomes", type=int, default=2); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
seqs = synth.genomes(a.L, a.genomes, seed=42)
idx = reveallib.index()
for s in seqs:
    idx.addsample("s"); idx.addsequence(s)
idx.set_option("RV_NO_CASCADE", 1)
idx.construct(); r = idx.align_builtin(20, 2)
idx.prof(enable=True, reset=True)
t0 = time.perf_counter()
for _ in range(a.steps):
    idx.construct(); r = idx.align_builtin(20, 2)
dt = (time.perf_counter() - t0) / a.steps
pl = idx.prof(enable=False)
l, off, pos = r["anchors"]
order = np.lexsort((pos[off[:-1]], l))
dig = hashlib.sha256(np.ascontiguousarray(l[order]).tobytes() + np.ascontiguousarray(pos[off[:-1]][order]).tobytes()).hexdigest()[:16]
print(json.dumps({"lib_dir": os.environ.get("RV_LIB_DIR", ""), "ms_per_step": dt * 1e3, "levels": r["stats"]["levels"], "anchors": int(len(l)), "anchor_digest": dig,
                  "classes_ms": {k: round(v[1] / a.steps, 2) for k, v in pl.items() if v[0] and v[1] / a.steps >= 0.05}}))
