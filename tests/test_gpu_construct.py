"""construct() and getmums() on the GPU, bit-exact against the CPU oracle
(reveallib/interface.c:160-291, reveallib/reveal.c:55-116)."""
import numpy as np
import pytest

from helpers import assemble, fa, feed, oracle, synth

pytestmark = pytest.mark.gpu

SETS = {
    "known": ["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"],
    "t1t2": fa("t1", "t2"),
    "1a1b": fa("1a", "1b"),
    "1a1b1c": fa("1a", "1b", "1c"),
    "1e1b": fa("1e", "1b"),
    "d1d2": fa("d1", "d2"),
    "1a1a": fa("1a", "1a"),
    "5way": fa("1a", "1b", "1c", "1d", "1e"),
}


def mod(sa64):
    from reveal_amd import reveallib, reveallib64
    return reveallib64 if sa64 else reveallib


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("name", list(SETS))
def test_construct_matches_oracle(name, sa64):
    T, nsep, nodes = assemble(SETS[name])
    O = oracle(sa64)
    c = O.construct(T, nsep, len(SETS[name]))
    idx = feed(mod(sa64).index(), SETS[name])
    assert idx.n == len(T) and idx.nsep == nsep and sorted(idx.nodes) == nodes
    idx.construct()
    assert idx.T.encode("latin-1") == T
    assert np.array_equal(idx.array("SA"), c["SA"])
    assert np.array_equal(idx.array("SAi"), c["SAi"])
    assert np.array_equal(idx.array("LCP"), c["LCP"])
    if len(SETS[name]) > 2:
        assert np.array_equal(idx.array("SO"), c["SO"])
    for minl in (1, 20):
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, minl)
        assert idx.getmums(minl) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def test_known_answer_vectors():
    """SURVEY.md 8(c) / reveal/tests/test_reveal.py:37 input"""
    idx = feed(mod(False).index(), SETS["known"])
    idx.construct()
    assert idx.T == "ACTTGCTAGCTAGTCAG$ACTAGCTAGCTAGTGAG$"
    assert idx.SA == [35, 17, 18, 0, 33, 15, 21, 7, 25, 11, 29, 14, 19, 5, 23, 9, 27, 1, 34, 16, 32, 4, 22, 8, 26, 12, 30, 20, 6, 24, 10, 28, 13, 31, 3, 2]
    assert idx.LCP == [0, 0, 0, 3, 1, 2, 2, 6, 7, 2, 3, 0, 1, 8, 9, 4, 5, 2, 0, 1, 1, 1, 10, 5, 6, 1, 2, 0, 7, 8, 3, 4, 1, 1, 2, 1]
    assert idx.getmums(1) == [(3, (0, 18), 0), (10, (4, 22), 0), (2, (3, 31), 0)]


@pytest.mark.parametrize("L,cnt", [(1000, 2), (50000, 2), (300000, 3)])
def test_construct_synthetic(L, cnt):
    seqs = [g.decode() for g in synth.genomes(L, cnt)]
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, cnt)
    idx = feed(mod(False).index(), seqs)
    idx.construct()
    assert np.array_equal(idx.array("SA"), c["SA"])
    assert np.array_equal(idx.array("LCP"), c["LCP"])
    l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 20)
    assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def test_construct_rc_and_files(tmp_path, monkeypatch):
    """construct(rc=1) remap (interface.c:168-175, reveal.c:98-100) and the sa=/lcp=/cache= files"""
    monkeypatch.chdir(tmp_path)
    T, nsep, nodes = assemble(fa("1a", "1brc"))
    O = oracle(False)
    tb = O.textbuf(T)
    O.revcomp(tb[nsep[0]:len(T)])
    SA = O.suffix_array(tb); SAi = O.inverse(SA); LCP = O.compute_lcp(tb, SA, SAi)
    l, a, b = O.getmums(tb, SA, LCP, nsep, 20, rc=1, nT=len(T))
    idx = feed(mod(False).index(cache=1), fa("1a", "1brc"))
    idx.construct(rc=1)
    assert np.array_equal(idx.array("SA"), SA) and np.array_equal(idx.array("LCP"), LCP)
    assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 1) for k in range(len(l))]
    assert (tmp_path / ".reveal.sa").stat().st_size == 4 * len(T)
    # read the cached arrays back instead of computing
    idx2 = feed(mod(False).index(sa=str(tmp_path / ".reveal.sa"), lcp=str(tmp_path / ".reveal.lcp")), fa("1a", "1brc"))
    idx2.construct(rc=1)
    assert np.array_equal(idx2.array("SA"), SA) and np.array_equal(idx2.array("LCP"), LCP)


def test_errors():
    m = mod(False)
    idx = m.index()
    with pytest.raises(m.error):
        idx.construct()                       # "No text to index."
    with pytest.raises(m.error):
        idx.addsample(3)
    idx.addsample("a"); idx.addsequence("ACGT")
    with pytest.raises(TypeError):
        idx.SA                                # "Index not yet constructed."
