#!/usr/bin/env python3
"""Where the replay of a run's anchors into the alignment graph spends its time (CPU only: the anchors come from a run of the Python
callbacks on the reference's own index, oracle/_ref/reveallib.so).  usage: python tools/time_replay.py [L=1000000] [genomes=5]"""
import os, sys, time, tempfile, pathlib, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from reveal_amd import rem, schemes, synth, alngraph
import pin_oracle as P

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
refmod = P.load_refmod(False)
tmp = pathlib.Path(tempfile.mkdtemp())
seqs = synth.genomes(L, K, seed=42, indelfrac=0.2)
files = []
for k, s in enumerate(seqs):
    p = tmp / ("g%d.fa" % k); p.write_text(">genome%d\n%s\n" % (k, s.decode())); files.append(str(p))
cache = pathlib.Path(ROOT) / "gpurun_out" / "replay" / ("anchors_%d_%d.pkl" % (L, K))
if cache.exists():
    anchors, gfa1, T_run = pickle.loads(cache.read_bytes())
else:
    rec = []
    class Rec(rem.GraphAligner):
        def graphalign(self, index, mum):
            rec.append((mum[0], mum[1], tuple(tuple(x) for x in mum[2])))
            return super().graphalign(index, mum)
    orig = rem.GraphAligner
    rem.GraphAligner = Rec
    t = time.perf_counter()
    G, idx, picker, aligner = rem.graph_align_genomes(files, indexmod=refmod, native=False)
    print("callbacks: %.1f s, %d anchors" % (time.perf_counter() - t, len(rec)))
    rem.GraphAligner = orig
    Tt = idx.T
    G.prune_nodes(Tt)
    alngraph.write_gfa(G, Tt, str(tmp / "a.gfa"), cmdline="x")
    anchors, gfa1, T_run = rec, (tmp / "a.gfa").read_bytes(), Tt
    cache.write_bytes(pickle.dumps((anchors, gfa1, T_run)))

def fresh():
    idx = refmod.index(); G = alngraph.AlnGraph()
    for f in files:
        alngraph.read_fasta(f, idx, G)
    return G, idx
# the text as the run left it: anchors lower-cased?  (write_gfa upper-cases aligned nodes; unaligned text is what the index holds)
import numpy as np
def canon(G):
    """the graph with its sentinels renamed (their names are random): everything the writer and prune_nodes look at, in dictionary order"""
    nm = {}
    for st in G.startnodes: nm[st] = ("start", tuple(G.offsets[st]))
    for en in G.endnodes: nm[en] = ("end", tuple(G.offsets[en]))
    f = lambda n: nm.get(n, n)
    return ([(f(n), list(o.items())) for n, o in G.offsets.items()], sorted((f(n), a) for n, a in G.aligned.items()),
            [(f(n), [((f(v), a, b), sorted(p)) for (v, a, b), p in d.items()]) for n, d in G.succ.items()],
            [(f(n), [((f(v), a, b), sorted(p)) for (v, a, b), p in d.items()]) for n, d in G.pred.items()])
def native(G, aligner, root_nodes, anchors):
    l = np.array([a[0] for a in anchors], dtype=np.uint32)
    off = np.zeros(len(anchors) + 1, dtype=np.int64); off[1:] = np.cumsum([len(a[2]) for a in anchors])
    pos = np.array([p for a in anchors for _, p in a[2]], dtype=np.int64)
    t = time.perf_counter()
    G.replay_native(root_nodes, l, off, pos)
    print("  (replay_native alone: %.2f s)" % (time.perf_counter() - t))
graphs = {}
for name, fn in (("replay_fast", rem.replay_anchors_fast), ("native", native)):
    G, idx = fresh()
    root_nodes = sorted(tuple(x) for x in idx.nodes)
    T = {}
    t = time.perf_counter()
    fn(G, rem.GraphAligner(G), root_nodes, anchors)
    T["replay"] = time.perf_counter() - t; t = time.perf_counter()
    graphs[name] = canon(G)
    t = time.perf_counter()
    Tt = T_run
    G.prune_nodes(Tt)
    T["prune"] = time.perf_counter() - t; t = time.perf_counter()
    alngraph.write_gfa(G, Tt, str(tmp / "b.gfa"), cmdline="x")
    T["write"] = time.perf_counter() - t
    same = (tmp / "b.gfa").read_bytes() == gfa1
    print(name, {k: round(v, 2) for k, v in T.items()}, "same GFA:", same, "nodes", G.number_of_nodes())
# everything behind the ABI: replay, prune_nodes, the GFA text
G, idx = fresh()
root_nodes = sorted(tuple(x) for x in idx.nodes)
l = np.array([a[0] for a in anchors], dtype=np.uint32)
off = np.zeros(len(anchors) + 1, dtype=np.int64); off[1:] = np.cumsum([len(a[2]) for a in anchors])
pos = np.array([p for a in anchors for _, p in a[2]], dtype=np.int64)
T = {}
t = time.perf_counter()
ng = alngraph.NativeGraph(G, root_nodes, l, off, pos)
T["replay"] = time.perf_counter() - t; t = time.perf_counter()
Tb = T_run.encode("latin-1")
ng.prune(Tb)
T["prune"] = time.perf_counter() - t; t = time.perf_counter()
ng.write_gfa(Tb, str(tmp / "c.gfa"), cmdline="x")
T["write"] = time.perf_counter() - t; t = time.perf_counter()
ng.load_into(G)
T["load_into"] = time.perf_counter() - t
print("behind the ABI", {k: round(v, 2) for k, v in T.items()}, "same GFA:", (tmp / "c.gfa").read_bytes() == gfa1)
alngraph.write_gfa(G, T_run, str(tmp / "d.gfa"), cmdline="x")
print("  loaded back and written by Python: same GFA:", (tmp / "d.gfa").read_bytes() == gfa1)
a, b = graphs["replay_fast"], graphs["native"]
print("same structure before prune_nodes:", [x == y for x, y in zip(a, b)])
