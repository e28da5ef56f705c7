#!/usr/bin/env python3
"""host-side anatomy of a step at C4: python tools/step_anatomy3.py [L]  -- wall time of construct(), of the C call rv_align_builtin and of the result's way into numpy"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reveal_amd import reveallib, synth, _lib
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250_000_000
seqs = synth.genomes(L, 2, seed=42)
idx = reveallib.index()
for k, s in enumerate(seqs):
    idx.addsample("g%d" % k); idx.addsequence(s)
dll, h = idx._dll, idx._h
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.construct(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t1s = time.perf_counter()
    dll.rv_set_trace(h, 0); idx._offer_result_buffers()
    st = _lib.RvAlignStats(); t2 = time.perf_counter()
    rc = dll.rv_align_builtin(h, 20, 2, ctypes.byref(st)); t3 = time.perf_counter()
    res = idx._builtin_result(st, False); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    del res
    print("construct %.2f (+sync %.2f) | offer %.3f | rv_align_builtin %.2f | result %.3f | sync %.3f | total %.2f ms" % ((t1 - t0) * 1e3, (t1s - t1) * 1e3, (t2 - t1s) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3))
