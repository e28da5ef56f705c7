"""The alignment graph of a finished run built behind the ABI (rv_graph_replay / rv_graph_prune / rv_graph_gfa, reveal_amd/csrc/rv_graph.hip;
reveal_amd/alngraph.py NativeGraph) beside the Python graph layer doing the same surgery (reveal/rem.py:14-200, 318-345, 384-447;
utils.py:710-839): same nodes, links and path sets in the same dictionary order after the replay and after prune_nodes, the same GFA byte for
byte.  Host code only: the anchors come from runs of the Python callbacks on the REFERENCE's own index (oracle/_ref/reveallib.so)."""
import os
import random
import sys

import numpy as np
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


def canon(G):
    """the graph with its sentinels renamed (their names are random): everything the writer and prune_nodes look at, in dictionary order"""
    nm = {}
    for st in G.startnodes:
        nm[st] = ("start", tuple(G.offsets[st]))
    for en in G.endnodes:
        nm[en] = ("end", tuple(G.offsets[en]))
    f = lambda n: nm.get(n, n)
    links = lambda D: [(f(n), [((f(v), a, b), sorted(p)) for (v, a, b), p in d.items()]) for n, d in D.items()]
    return [(f(n), list(o.items())) for n, o in G.offsets.items()], sorted((f(n), a) for n, a in G.aligned.items()), links(G.succ), links(G.pred)


def run_and_record(files, refmod, **kw):
    """the callbacks' run -> (anchors in call order, text after the run, graph)"""
    rec = []

    class Rec(rem.GraphAligner):
        def graphalign(self, index, mum):
            rec.append((mum[0], tuple(p for _, p in mum[2])))
            return super().graphalign(index, mum)
    orig = rem.GraphAligner
    rem.GraphAligner = Rec
    try:
        G, idx, picker, aligner = rem.graph_align_genomes(files, indexmod=refmod, native=False, **kw)
    finally:
        rem.GraphAligner = orig
    return rec, idx.T, G


def fresh(files, refmod):
    idx, G = refmod.index(), alngraph.AlnGraph()
    for f in files:
        alngraph.read_fasta(f, idx, G)
    return G, sorted(tuple(x) for x in idx.nodes)


def arrays(rec):
    l = np.array([a[0] for a in rec], dtype=np.uint32)
    off = np.zeros(len(rec) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(a[1]) for a in rec])
    return l, off, np.array([p for a in rec for p in a[1]], dtype=np.int64)


def compare(files, refmod, tmp_path, **kw):
    rec, T, G0 = run_and_record(files, refmod, **kw)
    assert len(rec) > 0
    # the callbacks' own graph, pruned and written by Python: what everything below must reproduce
    want_replayed = canon(G0)
    G0.prune_nodes(T)
    want_pruned = canon(G0)
    alngraph.write_gfa(G0, T, str(tmp_path / "want.gfa"), cmdline="x")
    want = (tmp_path / "want.gfa").read_bytes()
    l, off, pos = arrays(rec)
    # (1) the surgery alone in Python (rem.replay_anchors_fast): graphalign's bookkeeping for the index leaves no trace in the graph
    G1, roots = fresh(files, refmod)
    rem.replay_anchors_fast(G1, rem.GraphAligner(G1), roots, [(a[0], len(a[1]), tuple((0, p) for p in a[1])) for a in rec])
    assert canon(G1) == want_replayed
    # (2) behind the ABI, loaded back after the replay
    G2, roots = fresh(files, refmod)
    G2.replay_native(roots, l, off, pos)
    assert canon(G2) == want_replayed
    G2.prune_nodes(T)
    assert canon(G2) == want_pruned
    # (3) replay, prune_nodes and the GFA text behind the ABI; loaded back after prune_nodes
    G3, roots = fresh(files, refmod)
    with alngraph.NativeGraph(G3, roots, l, off, pos) as ng:
        ng.prune(T)
        assert ng.gfa(T, cmdline="x") == want
        os.environ["RV_GFA_PARALLEL_MIN"] = "0"          # the writer's threaded form (stretches of nodes and paths side by side) on a graph this small
        try:
            assert ng.gfa(T, cmdline="x") == want
        finally:
            del os.environ["RV_GFA_PARALLEL_MIN"]
        assert ng.counts() == (len(G0.seq_nodes()), sum(len(d) for d in G0.succ.values()))
        ng.load_into(G3)
    assert canon(G3) == want_pruned
    alngraph.write_gfa(G3, T, str(tmp_path / "back.gfa"), cmdline="x")
    assert (tmp_path / "back.gfa").read_bytes() == want
    return len(rec)


@pytest.mark.parametrize("names", [("1a", "1b"), ("1a", "1b", "1c"), ("1a", "1f"), ("1a", "e2"), ("1a", "1b", "1c", "1d")])
def test_fixtures(tmp_path, refmod, names):
    compare(C.fasta_files(tmp_path, names), refmod, tmp_path)


def test_options(tmp_path, refmod):
    files = C.fasta_files(tmp_path, ("1a", "1b", "1c"))
    compare(files, refmod, tmp_path, args=schemes.PickerArgs(trim=False), preselect=False)
    compare(files, refmod, tmp_path, args=schemes.PickerArgs(seedsize=100), preselect=False)
    compare(files, refmod, tmp_path, minlength=10)


def test_random_families(tmp_path, refmod):
    """2-7 samples of a random base with SNPs, indels, identical copies and shared inserts: bubbles whose alleles repeat (prune_nodes' work),
    anchors that fit a node exactly, anchors of sample subsets"""
    rng = random.Random(11)
    total = 0
    for case in range(12):
        k = rng.choice([2, 3, 3, 4, 5, 7])
        L = rng.choice([2000, 8000, 30000])
        base = "".join(rng.choice("ACGT") for _ in range(L))
        alleles = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 40))) for _ in range(6)]
        seqs = []
        for s in range(k):
            out, i = [], 0
            while i < L:
                r = rng.random()
                if r < 0.004:
                    out.append(rng.choice("ACGT")); i += 1
                elif r < 0.006:
                    out.append(rng.choice(alleles)); i += rng.randint(0, 30)      # one of a few alleles: several samples share it
                elif r < 0.007:
                    i += rng.randint(1, 300)
                else:
                    out.append(base[i]); i += 1
            seqs.append("".join(out))
        if rng.random() < 0.3:
            seqs[-1] = seqs[0]
        files = []
        for s, q in enumerate(seqs):
            p = tmp_path / ("c%d_%d.fa" % (case, s))
            p.write_text(">s%d\n%s\n" % (s, q))
            files.append(str(p))
        sub = tmp_path / ("case%d" % case)
        sub.mkdir()
        total += compare(files, refmod, sub)
    assert total > 500


def test_anchor_outside_the_graph_is_refused(tmp_path, refmod):
    files = C.fasta_files(tmp_path, ("1a", "1b"))
    G, roots = fresh(files, refmod)
    with pytest.raises(RuntimeError, match="no node"):
        alngraph.NativeGraph(G, roots, np.array([30], np.uint32), np.array([0, 2], np.int64), np.array([10, roots[0][1]], np.int64))


def test_node_begins_list():
    """alngraph._Begins: the largest begin <= a position, under insertions, against bisect on a plain list"""
    import bisect
    rng = random.Random(3)
    for trial in range(60):
        B = alngraph._Begins(rng.sample(range(100000), rng.choice([0, 1, 5, 600, 3000])))
        ref = sorted(B)
        for _ in range(2000):
            if rng.random() < 0.5:
                x = rng.randrange(100000)
                if x not in ref:
                    B.add(x)
                    bisect.insort(ref, x)
            q = rng.randrange(-5, 100005)
            j = bisect.bisect_right(ref, q)
            assert B.pred(q) == (ref[j - 1] if j else None), (trial, q)
        assert list(B) == ref and len(B) == len(ref)


def test_csr_to_tuples():
    """_index._csr_to_tuples: the lists the scans return (reveal.c: (length, count, ((sample, position), ...)) per match) from the C ABI's arrays"""
    from reveal_amd._index import _csr_to_tuples
    got = _csr_to_tuples(2, np.array([5, 7], np.uint32), np.array([2, 3], np.int32), np.array([0, 2, 5], np.int64),
                         np.array([0, 1, 0, 1, 2, 9], np.uint16), np.array([10, 20, 30, 40, 50, 99], np.int64))
    assert got == [(5, 2, ((0, 10), (1, 20))), (7, 3, ((0, 30), (1, 40), (2, 50)))]
    assert _csr_to_tuples(0, np.zeros(1, np.uint32), np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.uint16), np.zeros(1, np.int64)) == []
