#!/usr/bin/env python3
"""construct() alone, a few repetitions: ms per call (GPU box).  usage: python tools/sa_probe.py [L] [genomes] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reveal_amd import reveallib, synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
idx = reveallib.index()
for k, s in enumerate(synth.genomes(L, G)):
    idx.addsample("g%d" % k)
    idx.addsequence(s)
idx.upload()
ts = []
for rep in range(R):
    t0 = time.perf_counter()
    idx.construct()
    ts.append((time.perf_counter() - t0) * 1e3)
print("construct ms:", " ".join("%.2f" % t for t in ts), "| env", {k: v for k, v in os.environ.items() if k.startswith("RV_")})
