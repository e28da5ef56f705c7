import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import pathlib, tempfile
import graphrem_cases as C
from reveal_amd import rem, schemes
tmp = pathlib.Path(tempfile.mkdtemp())
files = C.fasta_files(tmp, sys.argv[1:] or ["1a", "1b", "1c"])
picked = {}
for native in (False, True):
    calls = []
    if not native:
        orig = schemes.GraphPicker.graphmumpicker
        def wrap(self, mums, idx, precomputed=False, minlength=0, _o=orig, _c=calls):
            r = _o(self, mums, idx, precomputed=precomputed, minlength=minlength)
            _c.append((idx.depth, sorted(idx.nodes)[0][0], len(mums), precomputed, (r[0][0], r[0][2]) if r else None, len(r[1]) if r else 0, len(r[2]) if r else 0))
            return r
        schemes.GraphPicker.graphmumpicker = wrap
    G, idx, fn = rem.graph_rem(files, str(tmp / ("n%d.gfa" % native)), args=schemes.PickerArgs(), native=native, preselect=False)
    if not native:
        schemes.GraphPicker.graphmumpicker = orig
        picked[False] = [c for c in calls if c[4]]
        print("callbacks: calls", len(calls), "picked", len(picked[False]))
    else:
        print("native:", idx.picker_info())
    print(native, "nodes", len(G.seq_nodes()), "aligned", sum(1 for n in G.seq_nodes() if G.aligned[n]))
# anchors of the native run again
from reveal_amd import reveallib
idx = reveallib.index()
for f in files:
    rem.read_fasta(f, idx)
idx.construct(); idx.set_picker(schemes.PickerArgs())
l, off, pos = idx.align_builtin(20, 2)["anchors"]
nat = [(int(l[k]), tuple(int(x) for x in pos[off[k]:off[k+1]])) for k in range(len(l))]
cb = [(c[4][0], tuple(p for _, p in c[4][1])) for c in picked[False]]
print("native anchors", len(nat), "callback anchors", len(cb))
sn, sc = set(nat), set(cb)
print("only native", sorted(sn - sc)[:5], "only callbacks", sorted(sc - sn)[:5])
for k, (a, b) in enumerate(zip(nat, cb)):
    if a != b:
        print("first difference at", k, a, b, picked[False][k]); break
