"""The two callbacks of `reveal rem` for GRAPH inputs behind the ABI (reveal_amd/csrc/rv_graphrem.hip: rv_graph_pick, rv_graph_align on the structure
rv_graph_import makes) beside their Python forms (schemes.GraphPicker.graphmumpicker, rem.GraphAligner.graphalign) on EVERY sub-index of whole
alignments: the reference's own index (oracle/_ref, no GPU) drives the recursion with the Python callbacks; a LoopGraph made from the same input graph
answers every call as well, and must answer the same -- the pick, the seeds, the children's intervals, the merged / left / right nodes -- and, node for
node and link for link in dictionary order, end up as the same graph, pruned and written to the same GFA text."""
import os
import sys

import pytest

import graphrem_cases as C
from reveal_amd import alngraph, rem, schemes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))


@pytest.fixture(scope="module")
def refmod():
    import pin_oracle as P
    mod = P.load_refmod(False)
    if mod is None:
        pytest.skip("oracle/_ref/reveallib.so not built (make -C oracle refmod needs /root/reference)")
    return mod


def beside(monkeypatch, stats):
    """every graphmumpicker / graphalign call of the run also goes to a LoopGraph of the same graph (made at the first call: the readers are done by then)"""
    state = {}
    py_pick, py_align = schemes.GraphPicker.graphmumpicker, rem.GraphAligner.graphalign

    def loop(G):
        if "lg" not in state:
            state["lg"] = alngraph.LoopGraph(G)
        return state["lg"]

    def pick(self, mums, idx, precomputed=False, minlength=0):
        lg = loop(self.G)
        if not precomputed and self.args.maxsize is None and self.args.maxdepth is None and len(mums):
            left, right = idx.leftnode, idx.rightnode
            got = lg.pick(list(mums), idx.nsamples, None if left is None else tuple(left), None if right is None else tuple(right), self.args, minlength)
        else:
            got = None
        want = py_pick(self, mums, idx, precomputed=precomputed, minlength=minlength)
        if got is not None:
            stats["picks"] += 1
            norm = lambda r: () if r == () else (r[0], [(tuple(m), s) for m, s in r[1]], [(tuple(m), s) for m, s in r[2]])
            w = () if want == () else ((want[0][0], want[0][1], tuple(tuple(x) for x in want[0][2])), [((m[0], m[1], tuple(tuple(x) for x in m[2])), s) for m, s in want[1]],
                                       [((m[0], m[1], tuple(tuple(x) for x in m[2])), s) for m, s in want[2]])
            assert norm(got) == norm(w), (got, w)
        return want

    def align(self, index, mum):
        lg = loop(self.G)
        nodes_before = sorted(tuple(x) for x in index.nodes)
        left, right = index.leftnode, index.rightnode
        got = lg.align(nodes_before, None if left is None else tuple(left), None if right is None else tuple(right), mum)
        want = py_align(self, index, mum)
        stats["aligns"] += 1
        leading, trailing, matching, rest, mn, newleft, newright = want
        node = lambda x: None if x is None else tuple(x)
        assert got == (sorted(tuple(x) for x in leading), sorted(tuple(x) for x in trailing), sorted(tuple(x) for x in matching), sorted(tuple(x) for x in rest),
                       node(mn), node(newleft), node(newright)), (got, want)
        return want
    monkeypatch.setattr(schemes.GraphPicker, "graphmumpicker", pick)
    monkeypatch.setattr(rem.GraphAligner, "graphalign", align)
    return state




@pytest.mark.parametrize("kw", [dict(), dict(trim=False), dict(seedsize=50), dict(gcmodel="star-avg", maxmums=50)])
def test_graph_inputs_native_beside_python(tmp_path, refmod, monkeypatch, kw):
    """level 0 in Python (two FASTA jobs), then the graph + graph job with both forms on every call; also a graph + FASTA job"""
    files = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d"])
    rem.graph_rem(files[:2], str(tmp_path / "ab.gfa"), indexmod=refmod, native=False)
    rem.graph_rem(files[2:], str(tmp_path / "cd.gfa"), indexmod=refmod, native=False)
    for inputs, name in (([str(tmp_path / "ab.gfa"), str(tmp_path / "cd.gfa")], "abcd.gfa"), ([str(tmp_path / "ab.gfa"), files[2]], "abc.gfa")):
        stats = dict(picks=0, aligns=0)
        with monkeypatch.context() as mp:
            state = beside(mp, stats)
            G, idx, picker, aligner = rem.graph_align_genomes(inputs, indexmod=refmod, native=False, args=schemes.PickerArgs(**kw), preselect=False)
        assert stats["aligns"] > 100 and stats["picks"] > 100
        lg = state["lg"]
        assert lg.snapshot() == alngraph.graph_snapshot(G)
        T = idx.T
        if len(G.paths) > 2:
            G.prune_nodes(T)
            lg.prune(T)
            assert lg.snapshot() == alngraph.graph_snapshot(G)
        fn = alngraph.write_gfa(G, T, str(tmp_path / ("py_" + name)), cmdline="x")
        assert lg.gfa(T, cmdline="x") == open(fn, "rb").read()
        lg.close()


def test_fasta_inputs_through_the_loop_graph(tmp_path, refmod, monkeypatch):
    """FASTA inputs (also a multi-contig one) are graphs, too: the same two entry points serve them"""
    files = C.fasta_files(tmp_path, ["1a", "1b", "1e"])
    stats = dict(picks=0, aligns=0)
    with monkeypatch.context() as mp:
        state = beside(mp, stats)
        G, idx, picker, aligner = rem.graph_align_genomes(files, indexmod=refmod, native=False, preselect=False)
    assert stats["aligns"] > 100
    assert state["lg"].snapshot() == alngraph.graph_snapshot(G)


def test_readers_behind_the_abi_leave_the_same_graph(tmp_path, refmod):
    """rv_graph_add_linear / rv_graph_read_gfa (csrc/rv_gfaread.hip) beside alngraph.read_fasta / read_gfa: graphs, FASTA files and both mixed, also a
    hand-made file with a segment and a link no path uses, two components, a path through one node twice and a '*' path -- node for node, link for link"""
    files = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d", "1e"])
    rem.graph_rem(files[:2], str(tmp_path / "ab.gfa"), indexmod=refmod, native=False)
    rem.graph_rem(files[2:4], str(tmp_path / "cd.gfa"), indexmod=refmod, native=False)
    rem.graph_rem([str(tmp_path / "ab.gfa"), str(tmp_path / "cd.gfa")], str(tmp_path / "abcd.gfa.gz"), indexmod=refmod, native=False)
    odd = tmp_path / "odd.gfa"
    odd.write_text("H\tVN:Z:1.0\n"
                   "S\t1\tACGTACGT\nS\t2\tggg\nS\t3\tTTTT\nS\t4\tCCCCC\nS\tx5\tAAAA\nS\t6\tGATTACA\nS\t7\tTT\n"
                   "L\t1\t+\t2\t+\t0M\nL\t1\t+\t3\t+\t0M\nL\t2\t+\t4\t+\t0M\nL\t3\t+\t4\t+\t0M\nL\t4\t+\t1\t+\t0M\nL\t3\t+\tx5\t+\t0M\nL\t6\t+\t7\t+\t0M\nL\t1\t+\t2\t+\t0M\n"
                   "P\tp1\t1+,2+,4+\t0M,0M\nP\tp2\t1+,3+,4+,1+\t0M,0M,0M\nP\t*p3\t6+,7+\t0M\nP\tp4\t6+\t\n")
    for inputs in ([str(tmp_path / "ab.gfa"), str(tmp_path / "cd.gfa")], [files[4], str(tmp_path / "abcd.gfa.gz"), files[0].replace("1a", "1a")], [str(odd)], [str(odd), files[4]]):
        if inputs[-1] == files[0]:
            inputs = inputs[:-1]
        Gp, tp = alngraph.AlnGraph(), C.TextOnly()
        for f in inputs:
            if f.endswith(".fa"):
                alngraph.read_fasta(f, tp, Gp)
            else:
                alngraph.read_gfa(f, tp, Gp)
        Gn, tn = alngraph.AlnGraph(), C.TextOnly()
        lg = alngraph.LoopGraph.read(inputs, tn, Gn)
        assert tn.n == tp.n
        assert Gn.paths == Gp.paths and Gn.id2end == Gp.id2end
        assert lg.snapshot() == alngraph.graph_snapshot(Gp)
        via_arrays = alngraph.LoopGraph(Gp)
        assert via_arrays.snapshot() == lg.snapshot()
        assert lg._dll.rv_graph_literal(lg._g) == (1 if Gp.literal_segments else 0)
        assert lg.counts() == via_arrays.counts() == (len(Gp.seq_nodes()), Gp.number_of_edges())
        # the one-call form of the reader (rv_graph_read_gfa) leaves the same graph as parse + adopt
        if all(not f.endswith(".fa") for f in inputs):
            import ctypes
            one = alngraph.LoopGraph.__new__(alngraph.LoopGraph)
            one._lib, one._dll = lg._lib, lg._dll
            one._g = lg._dll.rv_graph_new()
            tn = ctypes.c_int64(0)
            for f in inputs:
                import gzip
                data = (gzip.open if f.endswith(".gz") else open)(f, "rb").read()
                assert lg._dll.rv_graph_read_gfa(one._g, None, ctypes.byref(tn), data, len(data), None) > 0
            assert lg._dll.rv_graph_seal(one._g) == 0 and tn.value == tp.n
            one.names, one.sentinels0, one.nodes0 = list(Gp.paths), None, None
            assert one.snapshot() == lg.snapshot()
            one.close()
        # and back into Python objects
        n = lg.load_into(Gn)
        assert n == Gp.number_of_nodes() and alngraph.graph_snapshot(Gn) == alngraph.graph_snapshot(Gp)
        assert len(Gn.startnodes) == len(Gp.startnodes) and len(Gn.endnodes) == len(Gp.endnodes)
        lg.close(); via_arrays.close()


def test_reader_behind_the_abi_refuses_the_reverse_strand(tmp_path):
    bad = tmp_path / "rev.gfa"
    bad.write_text("S\t1\tACGT\nS\t2\tTTTT\nL\t1\t+\t2\t-\t0M\nP\tp\t1+,2-\t0M\n")
    with pytest.raises(alngraph.ReverseStrand):
        alngraph.LoopGraph.read([str(bad)], C.TextOnly(), alngraph.AlnGraph())
    broken = tmp_path / "broken.gfa"
    broken.write_text("S\t1\tACGT\nS\t2\tTTTT\nP\tp\t1+,2+\t0M\n")
    with pytest.raises(ValueError, match="link"):
        alngraph.LoopGraph.read([str(broken)], C.TextOnly(), alngraph.AlnGraph())


def test_literal_segmentgraph_beside_python(tmp_path, refmod, monkeypatch):
    """the reference's own form of segmentgraph (walks back from the end points, rem.py:282-287 / 303-308) forced on both sides: a graph + graph job call by call"""
    files = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d"])
    rem.graph_rem(files[:2], str(tmp_path / "ab.gfa"), indexmod=refmod, native=False)
    rem.graph_rem(files[2:], str(tmp_path / "cd.gfa"), indexmod=refmod, native=False)

    def forced(self):
        self.literal_segments = True
        return False
    monkeypatch.setattr(alngraph.AlnGraph, "check_segment_shortcut", forced)
    stats = dict(picks=0, aligns=0)
    with monkeypatch.context() as mp:
        state = beside(mp, stats)
        G, idx, picker, aligner = rem.graph_align_genomes([str(tmp_path / "ab.gfa"), str(tmp_path / "cd.gfa")], indexmod=refmod, native=False, preselect=False)
    assert G.literal_segments and stats["aligns"] > 100
    assert state["lg"]._dll.rv_graph_literal(state["lg"]._g) == 1
    assert state["lg"].snapshot() == alngraph.graph_snapshot(G)
