"""reveallib64 where it is needed: an index of more than 2^31 positions (reveallib/reveal.h:7-13, the guard of
interface.c:61-68 sends such inputs to the 64-bit module).  2 x 1.1 Gbp synthetic, n = 2 200 000 002 -- positions above
2^31 in SA, ranks above 2^31 in every level array -- through the size-independent properties of tests/test_gpu_fullsize.py:
SA a sorted permutation (checksums + sampled order), LCP Kasai-exact with the stops on sampled ranks incl. the top of the
array, the recursion's anchors exact / unique per sample / collinear / exactly the lower-cased bases, with the anchor cascade
and with the level pipeline (RV_NO_CASCADE), whose results must be identical.  And the cap: 2^32 - 2 positions and more are
refused by construct() with a message (ranks are carried in 32 bits inside the library, include/reveal_amd.h)."""
import numpy as np
import pytest

from helpers import synth
from reveal_amd import check

pytestmark = pytest.mark.gpu

L = 1_100_000_000


def _lcp_stop(T, a, b, cap=8192):
    n = len(T)
    h = 0
    while a + h < n and b + h < n and h < cap and T[a + h] == T[b + h] and T[b + h] not in (36, 78):
        h += 1
    return h


@pytest.fixture(scope="module")
def big():
    from reveal_amd import reveallib64
    seqs = synth.genomes(L, 2, seed=17)
    idx = reveallib64.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    T = np.frombuffer(b"$".join(seqs) + b"$", dtype=np.uint8)
    return idx, seqs, T


def test_construct_above_2_31(big):
    idx, seqs, T = big
    n = len(T)
    assert n == 2 * (L + 1) and n > 2 ** 31 and idx.n == n
    idx.construct()
    SA = idx.array("SA")
    assert SA.dtype == np.int64 and int(SA.min()) == 0 and int(SA.max()) == n - 1
    # a permutation of 0..n-1: sum and sum of squares modulo 2^64 (17.6 GB of positions: no second array of that size on the host)
    u = SA.view(np.uint64)
    assert int(u.sum(dtype=np.uint64)) == (n * (n - 1) // 2) % 2 ** 64
    with np.errstate(over="ignore"):
        sq = int((u * u).sum(dtype=np.uint64))
    assert sq == ((n - 1) * n * (2 * n - 1) // 6) % 2 ** 64
    LCP = idx.array("LCP")
    assert LCP[0] == 0 and int(LCP.max()) == idx.maxlcp
    rng = np.random.default_rng(3)
    ranks = np.concatenate([rng.integers(1, n, 3000), rng.integers(2 ** 31, n, 1500), np.arange(1, 200), np.arange(2 ** 31 - 100, 2 ** 31 + 100), np.arange(n - 200, n)])
    for k in ranks:
        a, b = int(SA[k - 1]), int(SA[k])
        h = _lcp_stop(T, a, b)
        assert LCP[k] == h, (k, a, b, LCP[k], h)             # compute_lcp, interface.c:97-114
        j = h
        while a + j < n and b + j < n and T[a + j] == T[b + j]:
            j += 1
        assert a + j >= n or (b + j < n and T[a + j] < T[b + j]), (k, a, b)
    assert int((SA[ranks] > 2 ** 31).sum()) > 100            # (positions that need the 64-bit module were among them)


def test_recursion_above_2_31(big, monkeypatch):
    idx, seqs, T0 = big
    nsep = np.cumsum([len(s) + 1 for s in seqs])[:-1] - 1
    out = []
    for off in (False, True):
        if off:
            idx.set_option("RV_NO_CASCADE", 1)      # (a switch of this handle: the library does not read the environment)
        idx.construct()
        res = idx.align_builtin(20, 2)
        assert idx.cascade_info()["done"] == (not off)
        l, o, pos = res["anchors"]
        assert res["stats"]["splits"] == len(l) and res["stats"]["anchored_bp"] == int(np.asarray(l, dtype=np.int64).sum())
        T1 = idx.array("T")
        if not off:
            p = check.recursion_properties(T0, T1, res["anchors"], nsep, 20)
            assert p["all"], p
            assert p["anchors"] > L // 150 and p["anchored_bp"] > 0.9 * L
            assert int(pos.max()) > 2 ** 31
        order = np.lexsort((pos[1::2], pos[0::2]))
        out.append((np.asarray(l)[order], pos[0::2][order], pos[1::2][order], T1, {k: res["stats"][k] for k in ("steps", "splits", "anchored_bp")}))
    a, b = out
    assert a[4] == b[4], (a[4], b[4])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3])


def test_the_cap_at_2_32_is_stated():
    """2^32 - 2 positions and more: refused by construct() in both modules (reveallib refuses the text at addsequence already,
    like the reference's int32 module would overflow: interface.c:61-68 raises there)"""
    from reveal_amd import reveallib64
    idx = reveallib64.index()
    chunk = synth.genomes(1 << 28, 1, seed=5)[0]
    idx.addsample("a")
    for _ in range(8):
        idx.addsequence(chunk)
    idx.addsample("b")
    for _ in range(8):
        idx.addsequence(chunk)
    assert idx.n >= 2 ** 32
    with pytest.raises(reveallib64.error, match="32 bits"):
        idx.construct()
