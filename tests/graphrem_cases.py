"""shared bodies of the `reveal rem` graph-driver tests: run with the reference's own index module on the CPU
(tests/test_cpu_graphrem.py, build container) and with reveal_amd's index on the GPU (tests/test_gpu_graphrem.py)"""
import gzip
import os

from helpers import GOLD
from reveal_amd import align, alngraph, rem, schemes


def fasta_files(tmp_path, names):
    out = []
    for x in names:
        dst = tmp_path / (x + ".fa")
        if not dst.exists():
            with gzip.open(os.path.join(GOLD, x + ".fa.gz"), "rt") as f:
                dst.write_text(f.read())
        out.append(str(dst))
    return out


def input_sequences(files):
    out = {}
    for f in files:
        for name, seq in rem.fasta_reader(f):
            out[name.replace(":", "").replace(";", "")] = seq.upper()
    return out


class TextOnly:
    """stands in for an index when a GFA is only read back to spell its paths"""
    def __init__(self):
        self.parts, self.n = [], 0

    def addsample(self, name):
        pass

    def addsequence(self, seq):
        b = self.n
        self.parts.append(seq + "$")
        self.n += len(seq) + 1
        return (b, self.n - 1)

    @property
    def T(self):
        return "".join(self.parts)


def spelled_by_file(gfa):
    G, t = alngraph.AlnGraph(), TextOnly()
    alngraph.read_gfa(gfa, t, G)
    T = t.T
    return {p: G.spell(p, T) for p in G.paths}, G


def check_graph(G, T, want):
    """test15's invariant (reveal/tests/test_reveal.py:150-159) for every path + what a merged node must be"""
    for name, seq in want.items():
        assert G.spell(name, T) == seq, name
    for n in G.seq_nodes():
        assert n[1] > n[0]
        if G.aligned[n]:
            assert len(G.offsets[n]) >= 2                 # an anchor joins at least two paths
        for sid, off in G.offsets[n].items():             # the node lies at its offset on each of its paths
            assert want[G.id2path[sid]][off:off + n[1] - n[0]] == T[n[0]:n[1]].upper()


def config1(tmp_path, indexmod):
    """BASELINE config 1 (`reveal rem tests/1a.fa tests/1b.fa`, defaults): the figures SURVEY.md section 6 / BASELINE.md record
    for the reference's own Python layer (553 anchors, 83 082 aligned bp, 1107 picker calls, 1646 nodes / 2197 edges incl. the
    four sentinels, 166 164 lower-case characters in the final text)"""
    files = fasta_files(tmp_path, ["1a", "1b"])
    G, idx, picker, aligner = rem.graph_align_genomes(files, indexmod=indexmod, preselect=False)
    T = idx.T
    al = [n for n in G.seq_nodes() if G.aligned[n]]
    assert len(al) == 553 and sum(e - b for b, e in al) == 83082
    assert picker.calls == 1107 and aligner.calls == 553
    assert len(G.offsets) == 1646 and sum(len(v) for v in G.succ.values()) == 2197
    assert sum(1 for c in T if c.islower()) == 166164
    check_graph(G, T, input_sequences(files))
    # the library-side pre-selection must not change anything when --trim is off
    args = schemes.PickerArgs(trim=False)
    Ga, ia, pa, aa = rem.graph_align_genomes(files, indexmod=indexmod, preselect=False, args=args)
    Gb, ib, pb, ab = rem.graph_align_genomes(files, indexmod=indexmod, preselect=True, args=args)
    assert sorted(n for n in Ga.seq_nodes() if Ga.aligned[n]) == sorted(n for n in Gb.seq_nodes() if Gb.aligned[n])
    assert ia.T == ib.T and pa.calls == pb.calls


def hierarchical(tmp_path, indexmod):
    """`reveal align --order=sequential --chunksize=2` over five fixture genomes: 2 + 1 + 1 + 1 jobs, graphs feeding graphs;
    the final GFA spells every input, both from the live graph and after reading the file back"""
    files = fasta_files(tmp_path, ["1a", "1b", "1c", "1d", "1e"])       # (1e is a multi-contig file: three paths)
    want = input_sequences(files)
    assert len(want) == 7
    levels = align.sequential_plan(files, 2, output=str(tmp_path / "prg"), tmpdir=str(tmp_path))
    assert [len(jobs) for jobs in levels] == [2, 1, 1]
    done = align.run_plan(levels, indexmod=indexmod)
    assert len(done) == 4
    final = levels[-1][-1][1]
    spelled, G = spelled_by_file(final)
    assert spelled == want
    assert sum(1 for n in G.seq_nodes() if len(G.offsets[n]) >= 4) > 50      # anchors shared by (nearly) all genomes survive the levels
    # the intermediate graphs too
    for jobs in levels[:-1]:
        for inputs, out in jobs:
            sp, _ = spelled_by_file(out)
            for name, seq in sp.items():
                assert want[name] == seq


def multi_fasta(tmp_path, indexmod):
    """three genomes at once (multi-MUM path, parallel children, prune_nodes) and a multi-contig input"""
    files = fasta_files(tmp_path, ["1a", "1b", "1c"])
    G, idx, fn = rem.graph_rem(files, str(tmp_path / "abc.gfa"), indexmod=indexmod)
    check_graph(G, idx.T, input_sequences(files))
    spelled, _ = spelled_by_file(fn)
    assert spelled == input_sequences(files)
    files = fasta_files(tmp_path, ["1e", "1b"])
    G, idx, fn = rem.graph_rem(files, str(tmp_path / "eb.gfa.gz"), indexmod=indexmod)
    assert fn.endswith(".gfa.gz")
    spelled, _ = spelled_by_file(fn)
    assert spelled == input_sequences(files)
