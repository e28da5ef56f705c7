"""Size-independent properties at a size the CPU oracle does not finish in seconds (2 x 20 Mbp, n = 4e7):
what construct() and the recursion must satisfy whatever the input (reveallib/interface.c:160-291, reveal.c:731-1338)."""
import numpy as np
import pytest

from helpers import synth

pytestmark = pytest.mark.gpu

L = 20_000_000


@pytest.fixture(scope="module")
def built():
    from reveal_amd import reveallib
    seqs = synth.genomes(L, 2, seed=7)
    idx = reveallib.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    idx.construct()
    T = np.frombuffer(idx.T.encode("latin-1"), dtype=np.uint8)
    SA = idx.array("SA").astype(np.int64)
    LCP = idx.array("LCP").astype(np.int64)
    return idx, seqs, T, SA, LCP


def _lcp_stop(T, a, b, cap=4096):
    """lcp of suffixes a, b with the reference's stops ('$' / 'N' at the later suffix, interface.c:107)"""
    n = len(T)
    h = 0
    while a + h < n and b + h < n and h < cap and T[a + h] == T[b + h] and T[b + h] not in (36, 78):
        h += 1
    return h


def test_sa_is_sorted_permutation_and_lcp_matches(built):
    idx, seqs, T, SA, LCP = built
    n = len(T)
    assert n == 2 * L + 2
    seen = np.zeros(n, dtype=np.uint8)
    seen[SA] = 1
    assert seen.all()                                        # a permutation of 0..n-1
    assert LCP[0] == 0
    rng = np.random.default_rng(3)
    ranks = np.concatenate([rng.integers(1, n, 4000), np.arange(1, 200), np.arange(n - 200, n)])
    for k in ranks:
        a, b = int(SA[k - 1]), int(SA[k])
        h = _lcp_stop(T, a, b)
        assert LCP[k] == h, (k, a, b, LCP[k], h)             # Kasai with stops
        # order: first differing byte decides; a suffix that is a prefix of the other sorts first
        x, y = T[a + h:a + h + 1], T[b + h:b + h + 1]
        if len(x) and len(y) and T[b + h] not in (36, 78):
            assert x[0] < y[0], (k, a, b)
        elif len(x) and len(y):                              # stopped at '$' / 'N': compare on
            j = h
            while a + j < n and b + j < n and T[a + j] == T[b + j]:
                j += 1
            assert a + j >= n or (b + j < n and T[a + j] < T[b + j]), (k, a, b)
    # the inverse
    SAi = idx.array("SAi").astype(np.int64)
    assert (SAi[SA[ranks]] == ranks).all()


def test_mums_are_maximal_unique_matches(built):
    idx, seqs, T, SA, LCP = built
    mums = idx.getmums(20)
    assert len(mums) > 100000
    rng = np.random.default_rng(5)
    sep = L                                                   # position of the first '$'
    for k in rng.integers(0, len(mums), 3000):
        l, (a, b), rc = mums[int(k)]
        assert rc == 0 and a < sep < b and l >= 20
        assert (T[a:a + l] == T[b:b + l]).all()               # a match
        assert T[a + l] != T[b + l] or T[a + l] in (36, 78)   # right-maximal
        assert a == 0 or T[a - 1] != T[b - 1] or T[a - 1] in (36, 78)      # left-maximal


def test_recursion_invariants(built):
    idx, seqs, T0, SA, LCP = built
    T0 = T0.copy()
    res = idx.align_builtin(20, 2, trace=False)
    l, off, pos = res["anchors"]
    l = np.asarray(l, dtype=np.int64); off = np.asarray(off); pos = np.asarray(pos, dtype=np.int64)
    assert (off[1:] - off[:-1] == 2).all()
    a, b = pos[0::2], pos[1::2]
    assert res["stats"]["splits"] == len(l) and res["stats"]["anchored_bp"] == int(l.sum())
    T1 = np.frombuffer(idx.T.encode("latin-1"), dtype=np.uint8)
    # test15's invariant (reveal/tests/test_reveal.py:150-159) in array form: the text still spells the input ...
    up = np.where((T1 >= 97) & (T1 <= 122), T1 - 32, T1)
    assert (up == T0).all()
    # ... exactly the anchored ranges are lower case, and anchors never overlap
    mask = np.zeros(len(T1) + 1, dtype=np.int64)
    for s in (a, b):
        np.add.at(mask, s, 1)
        np.add.at(mask, s + l, -1)
    cover = np.cumsum(mask)[:-1]
    assert cover.max() == 1
    assert (((T1 >= 97) & (T1 <= 122)) == (cover == 1)).all()
    # every anchor is an exact match of the two samples, long enough, collinear with all the others
    rng = np.random.default_rng(9)
    for k in rng.integers(0, len(l), 3000):
        assert l[k] >= 20 and (T0[a[k]:a[k] + l[k]] == T0[b[k]:b[k] + l[k]]).all()
    order = np.argsort(a)
    assert (np.diff(b[order]) > 0).all()                      # the linear interval model keeps both samples in order
    with pytest.raises(TypeError):
        idx.SA                                                # main SA/LCP are gone after align (reveal.c:1279-1284)
