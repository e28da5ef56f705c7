import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import pathlib, tempfile
from reveal_amd import rem, schemes, synth
tmp = pathlib.Path(tempfile.mkdtemp())
seqs = synth.genomes(200000, 5, seed=23, indelfrac=0.2)
files = []
for k, s in enumerate(seqs):
    p = tmp / ("g%d.fa" % k); p.write_text(">genome%d\n%s\n" % (k, s.decode())); files.append(str(p))
res = {}
orig = schemes.GraphPicker.graphmumpicker
calls = []
def wrap(self, mums, idx, precomputed=False, minlength=0):
    r = orig(self, mums, idx, precomputed=precomputed, minlength=minlength)
    if r:
        calls.append((r[0][0], tuple(sorted(p for _, p in r[0][2])), idx.depth, idx.nsamples, len(idx.nodes)))
    return r
schemes.GraphPicker.graphmumpicker = wrap
G, idx, fn = rem.graph_rem(files, str(tmp / "cb.gfa"), args=schemes.PickerArgs(), native=False, preselect=False)
schemes.GraphPicker.graphmumpicker = orig
print("callbacks: picked", len(calls), "nodes", len(G.seq_nodes()), "aligned", sum(1 for n in G.seq_nodes() if G.aligned[n]))
from reveal_amd import reveallib
ix = reveallib.index()
for f in files:
    rem.read_fasta(f, ix)
ix.construct(); ix.set_picker(schemes.PickerArgs())
r = ix.align_builtin(20, 2)
l, off, pos = r["anchors"]
nat = [(int(l[k]), tuple(sorted(int(x) for x in pos[off[k]:off[k+1]]))) for k in range(len(l))]
print("native anchors", len(nat), ix.picker_info())
cb = [(c[0], c[1]) for c in calls]
sn, sc = set(nat), set(cb)
print("only native", len(sn - sc), sorted(sn - sc)[:4]); print("only callbacks", len(sc - sn), sorted(sc - sn)[:4])
for c in calls:
    if (c[0], c[1]) in (sc - sn):
        print("  callback-only pick context: depth %d nsamples %d nodes %d members %d" % (c[2], c[3], c[4], len(c[1]))); break
