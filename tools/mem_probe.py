#!/usr/bin/env python3
"""Device memory a handle holds after construct + align_builtin (hipMemGetInfo through torch: the library binds to torch's runtime).
    python tools/mem_probe.py L [--sa64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reveal_amd import reveallib, reveallib64, synth
L = int(float(sys.argv[1])); sa64 = "--sa64" in sys.argv
free0, total = torch.cuda.mem_get_info()
seqs = synth.genomes(L, 2, seed=42)
idx = (reveallib64 if sa64 else reveallib).index()
for k, s in enumerate(seqs):
    idx.addsample("g%d" % k); idx.addsequence(s)
idx.construct()
torch.cuda.synchronize(); free1, _ = torch.cuda.mem_get_info()
idx.align_builtin(20, 2)
torch.cuda.synchronize(); free2, _ = torch.cuda.mem_get_info()
n = idx.n if hasattr(idx, "n") else 2 * (L + 1)
print("n = %d (%s-bit): device memory held after construct %.1f GB = %.1f B per position, after align %.1f GB = %.1f B per position; device total %.1f GB" % (
    n, 64 if sa64 else 32, (free0 - free1) / 1e9, (free0 - free1) / n, (free0 - free2) / 1e9, (free0 - free2) / n, total / 1e9))
