"""The Python-3 `reveal rem` graph driver (reveal_amd/rem.py graph_rem, alngraph.py, schemes.py, align.py) on the CPU.

The graph layer is host code; what it needs from an index it gets through the reference's own callback protocol, so in the
build container it can be driven by the REFERENCE's index -- its C sources built as the CPython module they define
(oracle/_ref/reveallib.so, `make -C oracle refmod`), real aligner() and all.  Without that module (no /root/reference at
build time) the index-driven tests are skipped; the graph-surgery unit tests and the plan need nothing."""
import os
import sys

import pytest

import graphrem_cases as C
from reveal_amd import align, alngraph

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))


@pytest.fixture(scope="module")
def refmod():
    import pin_oracle as P
    mod = P.load_refmod(False)
    if mod is None:
        pytest.skip("oracle/_ref/reveallib.so not built (make -C oracle refmod needs /root/reference)")
    return mod


def test_config1_figures_with_the_reference_index(tmp_path, refmod):
    C.config1(tmp_path, refmod)


def test_hierarchical_plan_with_the_reference_index(tmp_path, refmod):
    C.hierarchical(tmp_path, refmod)


def test_multi_and_multicontig_with_the_reference_index(tmp_path, refmod):
    C.multi_fasta(tmp_path, refmod)


def test_rem_align_entry_with_the_reference_index(refmod):
    """rem.align(aobjs, ...) -> (G, idx) (reveal/rem.py:616-712) driven by the reference's own index: test01 of
    reveal/tests/test_reveal.py:36-41, and the call shape of reveal/refine.py:220-229"""
    from reveal_amd import rem
    aobjs = [("1", "ACTTGCTAGCTAGTCAG"), ("2", "ACTAGCTAGCTAGTGAG")]
    G, idx = rem.align(aobjs, minlength=1, indexmod=refmod)
    assert G.number_of_nodes() > 2 and G.number_of_edges() > 2
    for name, seq in aobjs:
        assert G.spell_by_offsets(name, idx.T) == seq
    import random
    rng = random.Random(4)
    base = "".join(rng.choice("ACGT") for _ in range(700))
    var = list(base)
    for p in rng.sample(range(700), 12):
        var[p] = rng.choice("ACGT")
    aobjs = [("p0", base), ("p1", "".join(var).lower()), ("p2", base[:300] + base[340:]), ("gap", "")]
    G, idx = rem.align(aobjs, minlength=20, minn=2, seedsize=None, maxmums=1000, wpen=1, wscore=1, gcmodel="sumofpairs", sa64=False, indexmod=refmod)
    assert G.paths == ["p0", "p1", "p2"]
    for name, seq in aobjs[:3]:
        assert G.spell_by_offsets(name, idx.T) == seq.upper()
    assert sum(e - b for (b, e) in G.seq_nodes() if len(G.offsets[(b, e)]) == 3) > 400


def test_sequential_plan_is_align_py():
    """reveal/align.py:30-54"""
    plan = align.sequential_plan(["g%d.fa" % i for i in range(100)], 5, output="out")
    assert [len(jobs) for jobs in plan] == [20, 4, 1]                       # BASELINE config 5
    assert all(len(i) == 5 for i, _ in plan[0]) and all(len(i) == 5 for i, _ in plan[1]) and len(plan[2][0][0]) == 4
    assert plan[2][0][1] == "out.gfa"
    assert [len(j) for j in align.sequential_plan(list("abcdefg"), 2)] == [3, 2, 1]
    p = align.sequential_plan(list("abcdefg"), 2)
    assert p[1][0][0][0] == "g"                                               # the left-over input leads the next level
    assert [len(j) for j in align.sequential_plan(list("abcdef"), 2)] == [3, 1, 1]
    assert [len(j) for j in align.sequential_plan(list("abc"), 5)] == [1]


def _two_paths():
    """two paths over one node each: a = [0,100), b = [101,201)"""
    G = alngraph.AlnGraph()
    for sid, (name, iv) in enumerate((("a", (0, 100)), ("b", (101, 201)))):
        G.paths.append(name); G.path2id[name] = sid; G.id2path[sid] = name; G.id2end[sid] = 100
        s, e = "s%d" % sid, "e%d" % sid
        G.add_node(s, offsets={sid: 0}); G.startnodes.append(s)
        G.add_node(iv, offsets={sid: 0}, aligned=0)
        G.add_node(e, offsets={sid: 100}); G.endnodes.append(e)
        G.add_edge(s, iv, {sid}); G.add_edge(iv, e, {sid})
    return G


def test_breaknode_mergenodes_segmentgraph():
    """rem.py:14-131, 133-200, 260-316 on a graph small enough to check by eye"""
    G = _two_paths()
    assert G.node_at(0) == (0, 100) and G.node_at(150) == (101, 201)
    with pytest.raises(KeyError):
        G.node_at(100)                                                         # the '$' belongs to nobody
    m1, o1 = G.breaknode((0, 100), 40, 20)
    assert m1 == (40, 60) and o1 == {(0, 40), (60, 100)}
    assert G.offsets[(40, 60)] == {0: 40} and G.offsets[(60, 100)] == {0: 60} and G.offsets[(0, 40)] == {0: 0}
    assert G.node_at(39) == (0, 40) and G.node_at(40) == (40, 60) and G.node_at(99) == (60, 100)
    m2, o2 = G.breaknode((101, 201), 131, 20)
    assert m2 == (131, 151) and o2 == {(101, 131), (151, 201)}
    mn = G.mergenodes([m1, m2])
    assert mn == (40, 60) and G.aligned[mn] == 1 and G.offsets[mn] == {0: 40, 1: 30}
    assert not G.has_node((131, 151))
    assert set(v for (v, a, b) in G.succ[mn]) == {(60, 100), (151, 201)} and set(u for (u, a, b) in G.pred[mn]) == {(0, 40), (101, 131)}
    nodes = {(0, 40), (60, 100), (101, 131), (151, 201)}
    lead, trail, rest = G.segmentgraph(mn, nodes)
    assert lead == {(0, 40), (101, 131)} and trail == {(60, 100), (151, 201)} and rest == set()
    # a node that is not breakable: the whole node is the match
    m3, o3 = G.breaknode((0, 40), 0, 40)
    assert m3 == (0, 40) and o3 == set()
    # prefix-only / suffix-only cuts
    m4, o4 = G.breaknode((60, 100), 60, 10)
    assert m4 == (60, 70) and o4 == {(70, 100)} and G.node_at(60) == (60, 70) and G.node_at(70) == (70, 100)
    m5, o5 = G.breaknode((151, 201), 191, 10)
    assert m5 == (191, 201) and o5 == {(151, 191)}
    # paths still walk through everything
    T = "A" * 40 + "G" * 20 + "A" * 40 + "$" + "C" * 30 + "G" * 20 + "C" * 50 + "$"      # the merged stretch is the same 20 bases on both paths
    assert G.spell("a", T) == T[0:100] and G.spell("b", T) == T[101:201]


def test_segmentgraph_parallel_rest():
    """a third path that does not take part in the match: its node is neither in front nor behind -> rest"""
    G = _two_paths()
    G.paths.append("c"); G.path2id["c"] = 2; G.id2path[2] = "c"; G.id2end[2] = 50
    G.add_node("s2", offsets={2: 0}); G.startnodes.append("s2")
    G.add_node((202, 252), offsets={2: 0}, aligned=0)
    G.add_node("e2", offsets={2: 50}); G.endnodes.append("e2")
    G.add_edge("s2", (202, 252), {2}); G.add_edge((202, 252), "e2", {2})
    m1, _ = G.breaknode((0, 100), 10, 30)
    m2, _ = G.breaknode((101, 201), 111, 30)
    mn = G.mergenodes([m1, m2])
    nodes = {(0, 10), (40, 100), (101, 111), (141, 201), (202, 252)}
    lead, trail, rest = G.segmentgraph(mn, nodes)
    assert lead == {(0, 10), (101, 111)} and trail == {(40, 100), (141, 201)} and rest == {(202, 252)}


def test_gfa_round_trip_without_an_index(tmp_path):
    G = _two_paths()
    m1, _ = G.breaknode((0, 100), 40, 20)
    m2, _ = G.breaknode((101, 201), 141, 20)
    G.mergenodes([m1, m2])
    T = "ACGT" * 10 + "g" * 20 + "TTGA" * 10 + "$" + "CCAT" * 10 + "g" * 20 + "AAAC" * 10 + "$"
    fn = alngraph.write_gfa(G, T, str(tmp_path / "x.gfa"), cmdline="test")
    lines = open(fn).read().splitlines()
    assert lines[0] == "H\tVN:Z:1.0\tCL:Z:test"
    assert sum(1 for l in lines if l.startswith("S")) == 5 and sum(1 for l in lines if l.startswith("L")) == 4
    assert "S\t%d\t%s" % (1 + G.seq_nodes().index((40, 60)), "G" * 20) in lines            # aligned nodes are written upper case
    spelled, G2 = C.spelled_by_file(fn)
    assert spelled == {"a": T[0:100].upper(), "b": T[101:201].upper()}
    assert len(G2.startnodes) == 1 and len(G2.endnodes) == 1                                   # one component: sentinels merged


PLAN_WORKER = r"""
import os, sys, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle")); sys.path.insert(0, os.path.join(%r, "tests"))
import torch.distributed as dist
import pin_oracle as P
from reveal_amd import align
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
files = sorted(os.path.join(%r, f) for f in os.listdir(%r) if f.endswith(".fa"))
levels = align.sequential_plan(files, 2, output=os.path.join(%r, "prg"), tmpdir=%r)
done = align.run_plan(levels, rank=rank, world=world, barrier=dist.barrier, indexmod=P.load_refmod(False))
out = [None] * world
dist.all_gather_object(out, [(lv, j) for lv, j, *_ in done])
if rank == 0:
    print(json.dumps({"plan": [len(j) for j in levels], "jobs_by_rank": out, "final": levels[-1][-1][1]}))
dist.barrier(); dist.destroy_process_group()
"""


def test_plan_over_two_ranks_gloo(tmp_path, refmod):
    """N > 1 for config 5's shape: the jobs of a level are independent `reveal rem` runs (reveal/align.py:45-53) -- job j of a level
    belongs to rank j mod world, the GFA files are the only exchange, a barrier separates the levels.  Two gloo processes on the
    CPU (index = the reference's own module); the final graph spells every input."""
    import json
    import subprocess
    from helpers import ROOT
    files = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d", "1e"])
    want = C.input_sequences(files)
    script = tmp_path / "plan_worker.py"
    d = str(tmp_path)
    script.write_text(PLAN_WORKER % (ROOT, ROOT, ROOT, d, d, d, d))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert r["plan"] == [2, 1, 1]
    assert sorted(tuple(x) for x in r["jobs_by_rank"][0] + r["jobs_by_rank"][1]) == [(0, 0), (0, 1), (1, 0), (2, 0)]
    assert [tuple(x) for x in r["jobs_by_rank"][1]] == [(0, 1)]                 # rank 1 took job 1 of level 0
    spelled, _ = C.spelled_by_file(r["final"])
    assert spelled == want


def test_segmentgraph_without_the_walks_back_is_the_reference_form(tmp_path, refmod, monkeypatch):
    """every graphalign call of a pair, a three-way and a graph-of-graphs alignment: the walks back from the end points
    (rem.py:282-287, 303-308) never change leading / trailing / rest"""
    calls = [0]
    fast = alngraph.AlnGraph.segmentgraph

    def both(self, node, nodes):
        a = fast(self, node, nodes)
        b = self.segmentgraph_literal(node, nodes)
        assert a == b
        calls[0] += 1
        return a
    monkeypatch.setattr(alngraph.AlnGraph, "segmentgraph", both)
    from reveal_amd import rem
    files = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d"])
    rem.graph_rem(files[:2], str(tmp_path / "ab.gfa"), indexmod=refmod)
    rem.graph_rem(files[2:], str(tmp_path / "cd.gfa"), indexmod=refmod)
    rem.graph_rem(files[:3], str(tmp_path / "abc.gfa"), indexmod=refmod)
    rem.graph_rem([str(tmp_path / "ab.gfa"), str(tmp_path / "cd.gfa")], str(tmp_path / "abcd.gfa"), indexmod=refmod)
    assert calls[0] > 1500


def test_segment_shortcut_is_only_taken_where_it_holds():
    """segmentgraph leaves out the reference's walks back from the end points (rem.py:282-287, 303-308), which is only the same when
    every sequence node goes on over a real-path edge in both directions; a graph from a file with a node no path leaves gets the
    literal form (advisor finding, round 2)"""
    G = alngraph.AlnGraph()
    G.paths, G.path2id, G.id2path = ["p"], {"p": 0}, {0: "p"}
    G.add_node("start", offsets={0: 0}); G.add_node("end", offsets={0: 30})
    G.add_node((0, 10), offsets={0: 0}, aligned=0); G.add_node((11, 21), offsets={0: 10}, aligned=0)
    G.add_edge("start", (0, 10), {0}); G.add_edge((0, 10), (11, 21), {0}); G.add_edge((11, 21), "end", {0})
    assert G.check_segment_shortcut() and not G.literal_segments
    G.add_node((22, 30), offsets={0: 20}, aligned=0)          # a node nothing leads away from
    G.add_edge((11, 21), (22, 30), {0})
    assert not G.check_segment_shortcut() and G.literal_segments
    lead, trail, rest = G.segmentgraph((0, 10), [(11, 21), (22, 30)])
    assert (lead, trail, rest) == G.segmentgraph_literal((0, 10), [(11, 21), (22, 30)])


def _native_beside_python(monkeypatch, stats):
    """every call of the Python picker (schemes.GraphPicker.graphmumpicker, the not-precomputed branch) is also put to rv_pick_chain"""
    import bisect
    from reveal_amd import schemes
    orig = schemes.GraphPicker.graphmumpicker

    def both(self, mums, idx, precomputed=False, minlength=0):
        want = orig(self, mums, idx, precomputed=precomputed, minlength=minlength)
        if precomputed or len(mums) == 0 or self.args.maxsize is not None or self.args.maxdepth is not None:
            return want
        G = self.G
        rpaths = [p for p in G.paths if not p.startswith("*")]
        begins = getattr(self, "_seq_begin", None)
        if begins is None:                                  # the sequences of the index: one per sample (the graph's first nodes)
            begins = self._seq_begin = sorted(b for b, e in stats["root_nodes"])
            assert len(begins) == len(rpaths)
        ns = len(begins)
        ivb, ive = [-1] * ns, [-1] * ns
        for b, e in idx.nodes:
            s = bisect.bisect_right(begins, b) - 1
            assert ivb[s] < 0, "more than one interval of a sample: not the native picker's case"
            ivb[s], ive[s] = b, e
        got = schemes.native_pick([(mm[0], mm[1], tuple(mm[2])) for mm in mums], idx.nsamples, begins, ivb, ive, self.args, minlength)
        norm = lambda r: () if not r else ((r[0][0], r[0][1], tuple(tuple(x) for x in r[0][2])),
                                           [((m[0], m[1], tuple(tuple(x) for x in m[2])), sc) for m, sc in r[1]],
                                           [((m[0], m[1], tuple(tuple(x) for x in m[2])), sc) for m, sc in r[2]])
        assert norm(got) == norm(want), (idx.depth, sorted(idx.nodes), norm(got)[:1], norm(want)[:1])
        stats["calls"] += 1
        stats["picked"] += 1 if want else 0
        stats["seeded"] += 1 if want and (want[1] or want[2]) else 0
        return want
    monkeypatch.setattr(schemes.GraphPicker, "graphmumpicker", both)


@pytest.mark.parametrize("names,kw", [(["1a", "1b"], {}), (["1a", "1b", "1c"], {}), (["1a", "1b"], {"trim": False}), (["1a", "1c", "1d"], {"seedsize": 300, "maxmums": 50}),
                                      (["1a", "1b"], {"seedsize": 200, "wpen": 3, "gcmodel": "star-avg"}), (["d1", "d2"], {})])
def test_native_picker_decides_like_the_python_picker(tmp_path, refmod, monkeypatch, names, kw):
    """rv_pick_chain (C++ behind the ABI) beside schemes.GraphPicker.graphmumpicker on every sub-index of whole `reveal rem` runs driven by the
    reference's index: same choice (after trimming: length and positions), same seeds with the same scores, same refusals"""
    from reveal_amd import rem, schemes
    files = C.fasta_files(tmp_path, names)
    stats = {"calls": 0, "picked": 0, "seeded": 0, "root_nodes": None}
    orig_init = schemes.GraphPicker.__init__

    def init(self, graph, args=None):
        orig_init(self, graph, args)
        stats["root_nodes"] = list(graph.seq_nodes())
    monkeypatch.setattr(schemes.GraphPicker, "__init__", init)
    _native_beside_python(monkeypatch, stats)
    G, idx, fn = rem.graph_rem(files, str(tmp_path / "o.gfa"), args=schemes.PickerArgs(**kw), indexmod=refmod, preselect=False)
    if kw.get("seedsize", 10000) < 1000:      # (seeded children are "precomputed" calls: few calls reach the chain)
        assert stats["seeded"] > 0 and stats["calls"] >= 3, stats
    else:
        assert stats["calls"] > 5 and stats["picked"] > 5, stats
