#!/usr/bin/env python3
"""Time the graph code behind the ABI (rv_graph_replay / prune / gfa: host C++, no GPU needed) on the anchors tools/dump_anchors.py saved:
python tools/time_graph.py ANCHORS.npz"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reveal_amd import alngraph, shard, synth
d = np.load(sys.argv[1])
l, off, pos, L, K = d["l"], d["off"], d["pos"], int(d["L"]), int(d["K"])
seqs = synth.genomes(L, K, seed=42)
T0 = b"$".join(seqs) + b"$"
T = shard.lower_text(T0, (l, off, pos)).tobytes()


class FakeIndex:      # what alngraph.read_fasta asks of an index: addsample / addsequence -> (begin, end)
    def __init__(self):
        self.n = 0; self.nodes = []
    def addsample(self, name):
        pass
    def addsequence(self, s):
        b = self.n; self.n += len(s) + 1; self.nodes.append((b, self.n - 1)); return (b, self.n - 1)


tmp = tempfile.mkdtemp()
G = alngraph.AlnGraph(); idx = FakeIndex()
for k, s in enumerate(seqs):
    p = os.path.join(tmp, "g%d.fa" % k)
    open(p, "w").write(">genome%d\n%s\n" % (k, s.decode()))
    alngraph.read_fasta(p, idx, G)
root_nodes = sorted(idx.nodes)
t = time.perf_counter(); ng = alngraph.NativeGraph(G, root_nodes, l, off, pos); t1 = time.perf_counter()
ng.prune(T); t2 = time.perf_counter()
data = ng.gfa(T, cmdline="x"); t3 = time.perf_counter()
open(os.path.join(tmp, "o.gfa"), "wb").write(data); t4 = time.perf_counter()
import hashlib
print("replay %.3f s, prune %.3f s, gfa text %.3f s (%d bytes), file %.3f s; nodes, links = %s; sha %s" % (t1 - t, t2 - t1, t3 - t2, len(data), t4 - t3, ng.counts(), hashlib.sha256(data).hexdigest()[:16]))
