import sys; sys.path.insert(0,'.')
from reveal_amd import reveallib, synth
seqs = synth.genomes(5_000_000, 2, seed=42)
idx = reveallib.index()
for g in seqs: idx.addsample("s"); idx.addsequence(g.decode())
for i in range(2):
    idx.construct(); idx.align_builtin(20,2)
