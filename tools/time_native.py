#!/usr/bin/env python3
"""Where a `reveal rem` job of five 5 Mbp genomes with the native picker (rv_set_picker) spends its time: reading, construct, the recursion in the
library, the replay of the anchors into the alignment graph, prune_nodes, writing the GFA.  usage (GPU box): python tools/time_native.py"""
import sys, os, time, tempfile, pathlib, bisect
sys.path.insert(0, os.getcwd())
from reveal_amd import rem, schemes, synth, alngraph, reveallib
tmp = pathlib.Path(tempfile.mkdtemp())
seqs = synth.genomes(5000000, 5, seed=42)
files = []
for k, s in enumerate(seqs):
    p = tmp / ("g%d.fa" % k); p.write_text(">genome%d\n%s\n" % (k, s.decode())); files.append(str(p))
T = {}
t = time.perf_counter()
idx = reveallib.index(); G = alngraph.AlnGraph()
for f in files:
    alngraph.read_fasta(f, idx, G)
T["read"] = time.perf_counter() - t; t = time.perf_counter()
args = schemes.PickerArgs()
aligner = rem.GraphAligner(G)
root_nodes = sorted(tuple(x) for x in idx.nodes)
idx.construct(); idx.set_picker(args)
T["construct"] = time.perf_counter() - t; t = time.perf_counter()
res = idx.align_builtin(20, 2)
T["align_builtin"] = time.perf_counter() - t; t = time.perf_counter()
l, off, pos = res["anchors"]
begins = [b for b, _ in root_nodes]
pos = pos.tolist(); off = off.tolist(); l = l.tolist()
anchors = [(l[k], off[k + 1] - off[k], tuple((bisect.bisect_right(begins, p) - 1, p) for p in pos[off[k]:off[k + 1]])) for k in range(len(l))]
T["anchors_py"] = time.perf_counter() - t; t = time.perf_counter()
rem.replay_anchors(G, aligner, root_nodes, anchors)
T["replay"] = time.perf_counter() - t; t = time.perf_counter()
Tt = idx.T
G.prune_nodes(Tt)
T["prune"] = time.perf_counter() - t; t = time.perf_counter()
alngraph.write_gfa(G, Tt, str(tmp / "x.gfa"))
T["write"] = time.perf_counter() - t
print({k: round(v, 2) for k, v in T.items()}, len(anchors), idx.picker_info(), res["stats"])
