#!/usr/bin/env python3
"""Where a `reveal rem` job of five 5 Mbp genomes with the native picker (rv_set_picker) spends its time: reading, construct, the recursion in the
library, the anchors' surgery / prune_nodes / the GFA text behind the ABI (alngraph.NativeGraph), and -- for comparison -- the same graph work in
Python (rem.replay_anchors_fast, AlnGraph.prune_nodes, write_gfa).  usage (GPU box): python tools/time_native.py [L=5000000] [genomes=5] [--python]"""
import sys, os, time, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reveal_amd import rem, schemes, synth, alngraph, reveallib
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
L = int(argv[0]) if argv else 5000000
K = int(argv[1]) if len(argv) > 1 else 5
tmp = pathlib.Path(tempfile.mkdtemp())
seqs = synth.genomes(L, K, seed=42)
files = []
for k, s in enumerate(seqs):
    p = tmp / ("g%d.fa" % k); p.write_text(">genome%d\n%s\n" % (k, s.decode())); files.append(str(p))
# the whole job as the CLI runs it
t = time.perf_counter()
summary, idx0, fn = rem.graph_rem(files, str(tmp / "job.gfa"), materialize=False)
print("graph_rem(materialize=False): %.2f s  %s" % (time.perf_counter() - t, {k: (v if k != "paths" else len(v)) for k, v in summary.items()}))
del idx0
# ... and step by step
T = {}
t = time.perf_counter()
idx = reveallib.index(); G = alngraph.AlnGraph()
for f in files:
    alngraph.read_fasta(f, idx, G)
T["read"] = time.perf_counter() - t; t = time.perf_counter()
root_nodes = sorted(tuple(x) for x in idx.nodes)
idx.construct(); idx.set_picker(schemes.PickerArgs())
T["construct"] = time.perf_counter() - t; t = time.perf_counter()
res = idx.align_builtin(20, 2)
T["align_builtin"] = time.perf_counter() - t; t = time.perf_counter()
l, off, pos = res["anchors"]
Tt = idx.T
Tb = Tt.encode("latin-1")
T["text"] = time.perf_counter() - t; t = time.perf_counter()
ng = alngraph.NativeGraph(G, root_nodes, l, off, pos)
T["replay"] = time.perf_counter() - t; t = time.perf_counter()
ng.prune(Tb)
T["prune"] = time.perf_counter() - t; t = time.perf_counter()
ng.write_gfa(Tb, str(tmp / "native.gfa"), cmdline="x")
T["write"] = time.perf_counter() - t; t = time.perf_counter()
print("behind the ABI:", {k: round(v, 2) for k, v in T.items()}, len(l), "anchors", idx.picker_info(), res["stats"])
if "--python" in sys.argv:
    P = {}
    t = time.perf_counter()
    posl, offl, ll = pos.tolist(), off.tolist(), l.tolist()
    anchors = [(ll[k], offl[k + 1] - offl[k], tuple((0, p) for p in posl[offl[k]:offl[k + 1]])) for k in range(len(ll))]
    P["anchors_py"] = time.perf_counter() - t; t = time.perf_counter()
    rem.replay_anchors_fast(G, rem.GraphAligner(G), root_nodes, anchors)
    P["replay"] = time.perf_counter() - t; t = time.perf_counter()
    G.prune_nodes(Tt)
    P["prune"] = time.perf_counter() - t; t = time.perf_counter()
    alngraph.write_gfa(G, Tt, str(tmp / "python.gfa"), cmdline="x")
    P["write"] = time.perf_counter() - t; t = time.perf_counter()
    ng.load_into(alngraph.AlnGraph.__new__(alngraph.AlnGraph)) if False else None
    print("the same in Python:", {k: round(v, 2) for k, v in P.items()}, "same GFA:", (tmp / "python.gfa").read_bytes() == (tmp / "native.gfa").read_bytes())
ng.close()
