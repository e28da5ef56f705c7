"""The Python-3 `reveal rem` graph driver on top of reveal_amd's own index (HIP path): same bodies as
tests/test_cpu_graphrem.py runs with the reference's index -- BASELINE config 1's figures, graphs feeding graphs level by
level (reveal/align.py:27-54), multi-genome and multi-contig inputs -- plus a synthetic hierarchical run."""
import os

import numpy as np
import pytest

import graphrem_cases as C
from helpers import synth
from reveal_amd import align, rem

pytestmark = pytest.mark.gpu


def test_config1_figures(tmp_path):
    C.config1(tmp_path, None)


def test_hierarchical_plan(tmp_path):
    C.hierarchical(tmp_path, None)


def test_multi_and_multicontig(tmp_path):
    C.multi_fasta(tmp_path, None)


def test_64bit_library(tmp_path):
    files = C.fasta_files(tmp_path, ["1a", "1b"])
    G, idx, fn = rem.graph_rem(files, str(tmp_path / "ab64.gfa"), sa64=True)
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == C.input_sequences(files)
    assert sum(1 for n in G.seq_nodes() if G.aligned[n]) == 553


def test_synthetic_config5_shape(tmp_path):
    """BASELINE config 5 in miniature: 9 genomes of 60 kbp (1 % substitutions each), --order=sequential --chunksize=3:
    3 + 1 jobs; the final graph spells all nine"""
    seqs = synth.genomes(60000, 9, seed=21)
    files = []
    for k, s in enumerate(seqs):
        p = tmp_path / ("g%d.fa" % k)
        p.write_text(">genome%d\n%s\n" % (k, s.decode()))
        files.append(str(p))
    levels = align.sequential_plan(files, 3, output=str(tmp_path / "prg"), tmpdir=str(tmp_path))
    assert [len(j) for j in levels] == [3, 1]
    done = align.run_plan(levels)
    assert len(done) == 4
    spelled, G = C.spelled_by_file(levels[-1][-1][1])
    assert spelled == {"genome%d" % k: s.decode() for k, s in enumerate(seqs)}
    shared = [n for n in G.seq_nodes() if len(G.offsets[n]) == 9]
    assert sum(e - b for b, e in shared) > 0.3 * 60000


def test_config5_all_three_levels(tmp_path):
    """BASELINE config 5's own shape -- 100 genomes, --order=sequential --chunksize=5: 20 / 4 / 1 jobs, graphs feeding graphs twice -- at 1 Mbp
    per genome (tools/config5.py; the same command at 5 Mbp per genome is profiles/r06_config5_full_native.json): the final file spells all 100.
    Levels 1 and 2 run readers, picker and graphalign inside the library (rv_set_graph_picker)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("RV_")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "config5.py"), "--genomes", "100", "--L", "1000000", "--chunksize", "5", "--procs", "4", "--dir", str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["plan"] == [20, 4, 1] and d["paths_spell_inputs"] is True
    assert d["levels"]["0"]["jobs"] == 20 and d["levels"]["1"]["jobs"] == 4 and d["levels"]["2"]["jobs"] == 1


def _canonical_gfa(text):
    """a GFA file up to the numbering of its segments: the multiset of segment sequences, the links as pairs of sequences, every path as the list of
    the sequences it walks.  (Which of its members' intervals stands for a merged node -- and with it the node's number -- follows the order
    graphalign is handed the members in, and the order of equal truncated suffixes at a cut follows the iteration order of the Python set
    `matching`, reveal.c:673-674: neither is part of the alignment.)"""
    seg, links, paths = {}, [], {}
    for line in text.splitlines():
        f = line.split("\t")
        if f[0] == "S":
            seg[f[1]] = f[2]
        elif f[0] == "L":
            links.append((f[1], f[2], f[3], f[4]))
        elif f[0] == "P":
            paths[f[1]] = [(x[:-1], x[-1]) for x in f[2].split(",")] if f[2] else []
    return (sorted(seg.values()), sorted((seg[a], sa, seg[b], sb) for a, sa, b, sb in links), {k: [(seg[n], o) for n, o in v] for k, v in paths.items()})


@pytest.mark.parametrize("names,kw", [(["1a", "1b"], {}), (["1a", "1b", "1c"], {}), (["1a", "1b"], {"trim": False}), (["1a", "1c", "1d"], {"seedsize": 300, "maxmums": 50}),
                                      (["1a", "1b"], {"seedsize": 200, "wpen": 3, "gcmodel": "star-avg"}), (["d1", "d2"], {}), (["2a", "2b"], {})])
def test_native_picker_writes_the_same_gfa(tmp_path, names, kw):
    """`reveal rem` with its default picker inside the library (rv_set_picker / rv_pick_chain, no Python call per sub-index; the graph replayed
    from the anchors) against the same run through the two Python callbacks: the same file byte for byte"""
    from reveal_amd import schemes
    files = C.fasta_files(tmp_path, names)
    outs = {}
    for native in (False, True):
        G, idx, fn = rem.graph_rem(files, str(tmp_path / ("n%d.gfa" % native)), args=schemes.PickerArgs(**kw), native=native, preselect=False)
        outs[native] = (open(fn).read(), sum(1 for n in G.seq_nodes() if G.aligned[n]), len(G.seq_nodes()))
        if native:
            info = idx.picker_info()
            assert info["calls"] > 3
    assert outs[True][1:] == outs[False][1:]
    assert _canonical_gfa(outs[True][0]) == _canonical_gfa(outs[False][0])
    assert outs[True][0] == outs[False][0]
    if names == ["1a", "1b"] and not kw:
        assert outs[True][1] == 553      # BASELINE config 1's figure
    spelled, _ = C.spelled_by_file(str(tmp_path / "n1.gfa"))
    assert spelled == C.input_sequences(files)


def test_native_picker_synthetic_five_way(tmp_path):
    """five related genomes of 200 kbp: native picker == Python callbacks (anchors in sub-indices that lack samples, `segment`)"""
    from reveal_amd import schemes
    seqs = synth.genomes(200000, 5, seed=23, indelfrac=0.2)
    files = []
    for k, s in enumerate(seqs):
        p = tmp_path / ("g%d.fa" % k)
        p.write_text(">genome%d\n%s\n" % (k, s.decode()))
        files.append(str(p))
    texts = []
    for native in (False, True):
        G, idx, fn = rem.graph_rem(files, str(tmp_path / ("s%d.gfa" % native)), args=schemes.PickerArgs(), native=native, preselect=False)
        texts.append(open(fn).read())
    assert _canonical_gfa(texts[0]) == _canonical_gfa(texts[1])
    assert texts[0] == texts[1]


def _both_ways(tmp_path, files, tag, sa64=False, **kw):
    """graph_rem with both callbacks inside the library (rv_set_graph_picker) and through Python: -> the two GFA texts, aligned node counts"""
    from reveal_amd import schemes
    outs = {}
    for native in (False, True):
        G, idx, fn = rem.graph_rem(files, str(tmp_path / ("%s_n%d.gfa" % (tag, native))), args=schemes.PickerArgs(**kw), native=native, preselect=False, sa64=sa64)
        outs[native] = (open(fn).read(), sum(1 for n in G.seq_nodes() if G.aligned[n]), len(G.seq_nodes()), fn)
        if native:
            assert idx.picker_info()["kind"] == 0 and idx.picker_info()["calls"] > 0      # (switched off again after the run; the calls were counted)
    assert outs[True][1:3] == outs[False][1:3]
    assert outs[True][0] == outs[False][0]
    return outs[True][3]


@pytest.mark.parametrize("kw", [{}, {"trim": False}, {"seedsize": 300, "maxmums": 50}, {"seedsize": 200, "wpen": 3, "gcmodel": "star-avg"}])
def test_graph_inputs_native_callbacks_write_the_same_gfa(tmp_path, kw):
    """graphs as inputs (what levels 1 and 2 of `reveal align --order=sequential` run): graph + graph, graph + FASTA, three graphs -- the picker and graphalign
    of the reference's Python, inside the library on the graph behind the ABI (rv_graphrem.hip), against the Python callbacks: the same file byte for byte"""
    fa = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d"])
    g_ab = rem.graph_rem(fa[:2], str(tmp_path / "ab.gfa"))[2]
    g_cd = rem.graph_rem(fa[2:], str(tmp_path / "cd.gfa"))[2]
    fn = _both_ways(tmp_path, [g_ab, g_cd], "gg", **kw)
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == C.input_sequences(fa)
    fn = _both_ways(tmp_path, [g_ab, fa[2]], "gf", **kw)
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == C.input_sequences(fa[:3])
    if not kw:
        fn = _both_ways(tmp_path, [fa[3], fn], "fg3")      # (a graph of a graph: level 2)
        spelled, _ = C.spelled_by_file(fn)
        assert spelled == C.input_sequences(fa)


def test_multi_sequence_samples_native_callbacks(tmp_path):
    """samples of several sequences (multi-FASTA, one sample per file: --nocontigs is off) take the same route"""
    files = C.fasta_files(tmp_path, ["d1", "d2"])
    both = tmp_path / "both.fa"
    both.write_text(open(files[0]).read() + open(C.fasta_files(tmp_path, ["2a"])[0]).read())
    other = tmp_path / "other.fa"
    other.write_text(open(files[1]).read() + open(C.fasta_files(tmp_path, ["2b"])[0]).read())
    _both_ways(tmp_path, [str(both), str(other)], "mf")


def test_graph_inputs_native_synthetic(tmp_path):
    """nine related genomes of 60 kbp in three graphs, then the three graphs in one: level 1 of config 5's plan both ways"""
    seqs = synth.genomes(60000, 9, seed=29, indelfrac=0.2)
    files = []
    for k, s in enumerate(seqs):
        p = tmp_path / ("g%d.fa" % k)
        p.write_text(">genome%d\n%s\n" % (k, s.decode()))
        files.append(str(p))
    graphs = [rem.graph_rem(files[3 * j:3 * j + 3], str(tmp_path / ("j%d.gfa" % j)))[2] for j in range(3)]
    fn = _both_ways(tmp_path, graphs, "lvl1")
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == {"genome%d" % k: s.decode() for k, s in enumerate(seqs)}


@pytest.mark.parametrize("seed,kw", [(3, dict(snp=0.02, indelfrac=0.3)), (5, dict(snp=0.01, repeats=0.05)), (7, dict(snp=0.03, indelfrac=0.2, repeats=0.02, nruns=3)),
                                     (11, dict(snp=0.005, indelfrac=0.5))])
def test_graph_inputs_native_random_families(tmp_path, seed, kw):
    """unfriendly families (indels, interspersed repeats and tandem arrays, runs of N): seven genomes of 30 kbp as graphs of 3 + 2 + 2, then the three graphs in one -- both
    callbacks inside the library against the Python callbacks, the same file; also graph + graph + FASTA"""
    seqs = synth.family(30000, 7, seed=seed, **kw)
    files = []
    for k, s in enumerate(seqs):
        p = tmp_path / ("f%d.fa" % k)
        p.write_text(">fam%d\n%s\n" % (k, s.decode()))
        files.append(str(p))
    g1 = rem.graph_rem(files[0:3], str(tmp_path / "g1.gfa"))[2]
    g2 = rem.graph_rem(files[3:5], str(tmp_path / "g2.gfa"))[2]
    g3 = rem.graph_rem(files[5:7], str(tmp_path / "g3.gfa"))[2]
    fn = _both_ways(tmp_path, [g1, g2, g3], "fam")
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == {"fam%d" % k: s.decode().upper() for k, s in enumerate(seqs)}
    _both_ways(tmp_path, [g1, g2, files[5]], "famf", seedsize=100, maxmums=200)


def test_graph_inputs_native_64bit_library(tmp_path):
    """the same through libreveal_amd64.so (reveallib64: the graph behind the ABI is made by the library that runs the recursion)"""
    fa = C.fasta_files(tmp_path, ["1a", "1b", "1c"])
    g_ab = rem.graph_rem(fa[:2], str(tmp_path / "ab64.gfa"), sa64=True)[2]
    fn = _both_ways(tmp_path, [g_ab, fa[2]], "gf64", sa64=True)
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == C.input_sequences(fa)
    assert open(fn).read() == open(_both_ways(tmp_path, [g_ab, fa[2]], "gf32")).read()


def test_readers_behind_the_abi_fill_the_index_as_the_python_readers_do(tmp_path):
    """rv_gfa_parse / rv_graph_adopt / rv_add_sequences put the segments into the index (text, separators, intervals, samples) exactly as alngraph.read_gfa + addsequence do"""
    from reveal_amd import alngraph, reveallib
    fa = C.fasta_files(tmp_path, ["1a", "1b", "1c", "1d", "1e"])
    g_ab = rem.graph_rem(fa[:2], str(tmp_path / "ab.gfa"))[2]
    g_de = rem.graph_rem(fa[3:], str(tmp_path / "de.gfa.gz"))[2]
    inputs = [g_ab, fa[2], g_de]
    ip, Gp = reveallib.index(), alngraph.AlnGraph()
    for f in inputs:
        if f.endswith(".fa"):
            alngraph.read_fasta(f, ip, Gp)
        else:
            ip.addsample(os.path.basename(f))
            alngraph.read_gfa(f, ip, Gp)
    il, Gl = reveallib.index(), alngraph.AlnGraph()
    lg = alngraph.LoopGraph.read(inputs, il, Gl)
    assert il.n == ip.n and il.nsamples == ip.nsamples and il.samples == ip.samples
    assert il.nodes == ip.nodes                      # (asked of the library: rv_node_list)
    assert il.T == ip.T and il.nsep == ip.nsep
    assert Gl.paths == Gp.paths and Gl.id2end == Gp.id2end
    assert lg.snapshot() == alngraph.graph_snapshot(Gp)
    lg.close()
