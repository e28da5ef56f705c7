#!/usr/bin/env python3
"""Where one bench step goes on the host side: construct(), the library's rv_align_builtin, fetching the result into numpy.
usage (GPU box): python tools/step_anatomy.py [L] [genomes]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reveal_amd import _lib, reveallib, synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = reveallib.index()
for k, s in enumerate(synth.genomes(L, G)):
    idx.addsample("g%d" % k)
    idx.addsequence(s)
idx.upload()
for rep in range(4):
    t0 = time.perf_counter()
    idx.construct()
    t1 = time.perf_counter()
    st = _lib.RvAlignStats()
    idx._dll.rv_set_trace(idx._h, 0)
    assert idx._dll.rv_align_builtin(idx._h, 20, 2, ctypes.byref(st)) == 0
    t2 = time.perf_counter()
    res = idx._builtin_result(st, False)
    t3 = time.perf_counter()
    print("rep %d: construct %.2f ms, rv_align_builtin %.2f ms, result -> numpy %.2f ms (%d anchors)" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(res["anchors"][0])))
