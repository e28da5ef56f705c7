#!/usr/bin/env python3
"""Where the host time of one C4 step goes: construct, the C call of align_builtin, and the pieces of _builtin_result."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reveal_amd import reveallib, synth, _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
seqs = synth.genomes(L, 2, seed=42)
idx = reveallib.index()
for k, s in enumerate(seqs):
    idx.addsample("g%d" % k); idx.addsequence(s)
idx.upload()
dll, h = idx._dll, idx._h
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.construct()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = _lib.RvAlignStats()
    dll.rv_set_trace(h, 0)
    assert dll.rv_align_builtin(h, 20, 2, ctypes.byref(st)) == 0
    t2 = time.perf_counter()
    mem = ctypes.c_int64(0)
    na = dll.rv_anchor_count(h, ctypes.byref(mem))
    l = np.empty(max(na, 1), dtype=np.uint32); off = np.empty(na + 1, dtype=np.int64); pos = np.empty(max(mem.value, 1), dtype=np.int64)
    off[0] = 0
    t3 = time.perf_counter()
    dll.rv_fetch_anchors(h, l.ctypes.data, off.ctypes.data, pos.ctypes.data)
    t4 = time.perf_counter()
    dll.rv_fetch_anchors(h, l.ctypes.data, off.ctypes.data, pos.ctypes.data)      # the same buffers again: pages already there
    t5 = time.perf_counter()
    print("construct %.2f  align_builtin (C) %.2f  np.empty %.3f  fetch (fresh pages) %.2f  fetch again (touched pages) %.2f  | anchors %d" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, na), flush=True)
