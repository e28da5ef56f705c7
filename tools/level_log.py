#!/usr/bin/env python3
"""Per-level wall times of the built-in recursion (RV_LEVEL_LOG=1 adds a sync per level).
usage: python tools/level_log.py [genomes] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reveal_amd import reveallib, synth

g = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
seqs = synth.genomes(L, g, seed=42)
idx = reveallib.index()
for k, s in enumerate(seqs):
    idx.addsample("s%d" % k); idx.addsequence(s)
for rep in range(3):
    idx.construct()
    if rep == 2:
        idx.set_option("RV_LEVEL_LOG", 1)      # the first runs are silent warm-ups (device buffers reach their final size)
    t = time.time(); idx.align_builtin(20, 2, trace=False); print("total %.2f ms" % ((time.time() - t) * 1e3), file=sys.stderr)
