#!/usr/bin/env python3
"""The anchors of a `reveal rem` job with the native picker (rv_set_picker), saved for timing the graph code behind the ABI (rv_graph_*: host C++) on a
machine without a GPU: python tools/dump_anchors.py OUT.npz [L=5000000] [genomes=5]   (tools/time_graph.py reads it; the inputs are synth.genomes(L, K, seed=42))"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reveal_amd import reveallib, schemes, synth
out = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
idx = reveallib.index()
for k, s in enumerate(synth.genomes(L, K, seed=42)):
    idx.addsample("genome%d" % k); idx.addsequence(s)
idx.construct(); idx.set_picker(schemes.PickerArgs())
l, off, pos = idx.align_builtin(20, 2)["anchors"]
np.savez_compressed(out, l=l, off=off, pos=pos, L=L, K=K)
print(out, len(l), "anchors", idx.picker_info())
