"""`rem.align(aobjs, ...)` -> (G, idx) (reveal/rem.py:616-712): the entry reveal/refine.py:220-229 calls for the sequences of a
bubble and reveal/tests/test_reveal.py:36-41 for two 17-mers -- on reveal_amd's index (HIP path)."""
import random

import pytest

import graphrem_cases as C
from reveal_amd import rem

pytestmark = pytest.mark.gpu


def check(G, idx, aobjs):
    T = idx.T
    for name, seq in aobjs:
        if seq:
            assert G.spell_by_offsets(name, T) == seq.upper(), name          # every path spells its input (test15's invariant)
    for n in G.seq_nodes():
        assert n[1] > n[0]
    # (what the recursion matched is lower case, reveal.c:1230-1234; prune_nodes merges equal unmatched siblings without touching the text)
    assert sum(e - b for (b, e) in G.seq_nodes() if G.aligned[(b, e)] and T[b:e].islower()) > 0


def test01_seqpair_align():
    """reveal/tests/test_reveal.py:36-41, literally"""
    aobjs = [("1", "ACTTGCTAGCTAGTCAG"), ("2", "ACTAGCTAGCTAGTGAG")]
    G, idx = rem.align(aobjs, minlength=1)
    assert G.number_of_nodes() > 2
    assert G.number_of_edges() > 2
    check(G, idx, aobjs)
    assert any(len(G.offsets[n]) == 2 for n in G.seq_nodes())                 # something is shared by both sequences


def test_refine_call_shape():
    """the keyword arguments reveal/refine.py:220-229 passes; sequences of a few hundred bases, one of them empty"""
    rng = random.Random(4)
    base = "".join(rng.choice("ACGT") for _ in range(700))
    var = list(base)
    for p in rng.sample(range(700), 12):
        var[p] = rng.choice("ACGT")
    aobjs = [("p0", base), ("p1", "".join(var).lower()), ("p2", base[:300] + base[340:]), ("gap", "")]
    G, idx = rem.align(aobjs, minlength=20, minn=2, seedsize=None, maxmums=1000, wpen=1, wscore=1, gcmodel="sumofpairs", sa64=False)
    assert G.paths == ["p0", "p1", "p2"]
    check(G, idx, aobjs)
    assert sum(e - b for (b, e) in G.seq_nodes() if len(G.offsets[(b, e)]) == 3) > 400


@pytest.mark.parametrize("sa64", [False, True])
def test_same_alignment_as_the_file_driver(tmp_path, sa64):
    """the sequences of 1a.fa / 1b.fa through rem.align and through graph_rem (FASTA files, a start / end sentinel per sequence):
    the same aligned nodes"""
    files = C.fasta_files(tmp_path, ["1a", "1b"])
    aobjs = [(name, seq) for f in files for name, seq in rem.fasta_reader(f)]
    G, idx = rem.align(aobjs, minlength=20, maxmums=1000, sa64=sa64)
    G2, idx2, _ = rem.graph_rem(files, None, sa64=sa64, contigs=False)
    check(G, idx, aobjs)
    assert sorted(n for n in G.seq_nodes() if G.aligned[n]) == sorted(n for n in G2.seq_nodes() if G2.aligned[n])
    assert idx.T == idx2.T
