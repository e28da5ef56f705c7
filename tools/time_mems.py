#!/usr/bin/env python3
"""getmultimems on k synthetic genomes of L bases: seconds per call (GPU box).  usage: python tools/time_mems.py [L=5000000] [k=10] [minl=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reveal_amd import reveallib, synth
sys_args = [a for a in sys.argv[1:] if not a.startswith("--")]
L = int(sys_args[0]) if len(sys_args) > 0 else 5000000
k = int(sys_args[1]) if len(sys_args) > 1 else 10
minl = int(sys_args[2]) if len(sys_args) > 2 else 20
idx = reveallib.index()
for i, g in enumerate(synth.genomes(L, k, seed=42)):
    idx.addsample("s%d" % i); idx.addsequence(g.decode())
idx.construct()
import ctypes
for rep in range(3):
    members = ctypes.c_int64(0)
    t = time.perf_counter()
    cnt = idx._dll.rv_getmultimums(idx._h, minl, 2, 1, ctypes.byref(members))
    dt = time.perf_counter() - t
    print("rv_getmultimums(mems=1): %d x %g Mbp (n = %d), minl %d: %.1f ms to the host's vectors, %d multi-MEMs, %d members" % (k, L / 1e6, k * (L + 1), minl, dt * 1e3, cnt, members.value))
if "--no-tuples" in sys.argv:
    sys.exit(0)
t = time.perf_counter()
r = idx.getmultimems(minlength=minl, minn=2)
print("index.getmultimems (the same + the result as Python tuples): %.2f s" % (time.perf_counter() - t))
