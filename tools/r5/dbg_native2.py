import sys, os, bisect
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import pathlib, tempfile
import graphrem_cases as C
from reveal_amd import rem, schemes
tmp = pathlib.Path(tempfile.mkdtemp())
files = C.fasta_files(tmp, sys.argv[1:] or ["1a", "1b", "1c"])
orig = schemes.GraphPicker.graphmumpicker
state = {"begins": None, "n": 0, "bad": 0}
def wrap(self, mums, idx, precomputed=False, minlength=0):
    r = orig(self, mums, idx, precomputed=precomputed, minlength=minlength)
    if precomputed or not mums:
        return r
    if state["begins"] is None:
        state["begins"] = sorted(b for b, e in self.G.seq_nodes())
    begins = state["begins"]; ns = len(begins)
    ivb, ive = [-1] * ns, [-1] * ns
    for b, e in idx.nodes:
        s = bisect.bisect_right(begins, b) - 1
        ivb[s], ive[s] = b, e
    got = schemes.native_pick([(m[0], m[1], tuple(m[2])) for m in mums], idx.nsamples, begins, ivb, ive, self.args, minlength)
    state["n"] += 1
    a = (r[0][0], tuple(tuple(x) for x in r[0][2])) if r else None
    b = (got[0][0], tuple(tuple(x) for x in got[0][2])) if got else None
    if a != b:
        state["bad"] += 1
        if state["bad"] < 4:
            print("DIFF depth", idx.depth, "nodes", sorted(idx.nodes), "python", a, "native", b, "nmums", len(mums))
            print("   mums with these positions:", [m for m in mums if r and set(p for _, p in m[2]) == set(p for _, p in r[0][2])][:4])
    return r
schemes.GraphPicker.graphmumpicker = wrap
G, idx, fn = rem.graph_rem(files, str(tmp / "x.gfa"), args=schemes.PickerArgs(), native=False, preselect=False)
print("calls compared", state["n"], "different", state["bad"])
