"""BASELINE.json's full-size configurations through size-independent properties (the CPU oracle needs minutes there):

  C4  2 x 250 Mbp, n = 5*10^8   -- 39-bit first keys / 5 radix passes, the chunked carry scan above 64 M ranks,
                                   int32 positions near 2^31, the LDS bubble kernels of levels with thousands of children
  C3  10 x 5 Mbp, n = 5*10^7    -- the multi-sample scan / picker / split path at its benchmark size

What must hold whatever the input (reveallib/interface.c:160-291 construct, reveal.c:731-1338 the recursion,
tests/test_reveal.py:150-159 the reference's round trip): SA is a sorted permutation, LCP is Kasai with the stops, SAi its
inverse; every anchor of the recursion is an exact match present once in each of its samples, anchors never overlap and are
collinear, exactly the anchored bases are lower case, and the text still spells the input (reveal_amd/check.py).
"""
import numpy as np
import pytest

from helpers import synth
from reveal_amd import check

pytestmark = pytest.mark.gpu

CONFIGS = {"C4": (250_000_000, 2, 11), "C3": (5_000_000, 10, 13)}


def golden(cfg):
    """the CPU path's digests at this size (tests/golden/fullsize.json: the reference's divsufsort + the restated recursion, run in
    the build container by oracle/gen_fullsize_golden.py)"""
    L, G, seed = CONFIGS[cfg]
    rec = check.golden_record(L, G, seed)
    assert rec is not None, "tests/golden/fullsize.json holds no record for %s" % cfg
    return rec


def _lcp_stop(T, a, b, cap=8192):
    n = len(T)
    h = 0
    while a + h < n and b + h < n and h < cap and T[a + h] == T[b + h] and T[b + h] not in (36, 78):
        h += 1
    return h


@pytest.fixture(scope="module", params=["C4", "C3"])
def built(request):
    from reveal_amd import reveallib
    L, G, seed = CONFIGS[request.param]
    seqs = synth.genomes(L, G, seed=seed)
    idx = reveallib.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    idx.construct()
    T = np.frombuffer(b"$".join(seqs) + b"$", dtype=np.uint8)
    return request.param, idx, seqs, T


def test_construct_properties(built):
    name, idx, seqs, T = built
    L, G, _ = CONFIGS[name]
    n = len(T)
    assert n == G * (L + 1) and idx.n == n
    assert np.array_equal(idx.array("T"), T)
    SA = idx.array("SA")
    seen = np.zeros(n, dtype=np.uint8)
    seen[SA] = 1
    assert seen.all()                                        # a permutation of 0..n-1
    del seen
    LCP = idx.array("LCP")
    assert LCP[0] == 0 and int(LCP.max()) == idx.maxlcp
    # bit-identical to the reference's suffix array (divsufsort) and to compute_lcp (interface.c:97-114) at full size
    g = check.compare_with_golden(golden(name), SA=SA, LCP=LCP)
    assert g["all"], g
    assert idx.maxlcp == golden(name)["maxlcp"]
    rng = np.random.default_rng(3)
    ranks = np.concatenate([rng.integers(1, n, 6000), np.arange(1, 300), np.arange(n - 300, n)])
    for k in ranks:
        a, b = int(SA[k - 1]), int(SA[k])
        h = _lcp_stop(T, a, b)
        assert LCP[k] == h, (k, a, b, LCP[k], h)             # compute_lcp, interface.c:97-114
        j = h                                                # order: go on past a stop, first differing byte decides
        while a + j < n and b + j < n and T[a + j] == T[b + j]:
            j += 1
        assert a + j >= n or (b + j < n and T[a + j] < T[b + j]), (k, a, b)
    # every group of equal LCP-block neighbours is in text order only where the reference's sorter says so: not asserted;
    # the inverse is
    SAi = idx.array("SAi")
    assert (SAi[SA[ranks]] == ranks).all()
    probe = rng.integers(0, n, 100000)
    assert (SA[SAi[probe]] == probe).all()


def test_top_level_scan_properties(built):
    name, idx, seqs, T = built
    L, G, _ = CONFIGS[name]
    rng = np.random.default_rng(5)
    if G == 2:
        mums = idx.getmums(20)
        assert len(mums) > L // 200
        sep = L
        for k in rng.integers(0, len(mums), 4000):
            l, (a, b), rc = mums[int(k)]
            assert rc == 0 and a < sep < b and l >= 20
            assert (T[a:a + l] == T[b:b + l]).all()
            assert T[a + l] != T[b + l] or T[a + l] in (36, 78)
            assert a == 0 or T[a - 1] != T[b - 1] or T[a - 1] in (36, 78)
    else:
        mums = idx.getmultimums(20, 2)
        assert len(mums) > L // 200
        for k in rng.integers(0, len(mums), 3000):
            l, cnt, members = mums[int(k)]
            assert l >= 20 and 2 <= cnt <= G and len(members) == cnt
            assert len({s for s, _ in members}) == cnt                     # once per sample (reveal.c:436-580)
            p0 = members[0][1]
            for s, p in members:
                assert s == p // (L + 1)
                assert (T[p:p + l] == T[p0:p0 + l]).all()
            right = {int(T[p + l]) for _, p in members}
            assert len(right) > 1 or right <= {36, 78}                    # right-maximal
            left = {int(T[p - 1]) if p else 36 for _, p in members}
            assert len(left) > 1 or left <= {36, 78}                      # left-maximal


def test_recursion_properties(built):
    name, idx, seqs, T0 = built
    L, G, _ = CONFIGS[name]
    res = idx.align_builtin(20, 2, trace=False)
    l, off, pos = res["anchors"]
    assert res["stats"]["splits"] == len(l) and res["stats"]["anchored_bp"] == int(np.asarray(l, dtype=np.int64).sum())
    nsep = np.cumsum([len(s) + 1 for s in seqs])[:-1] - 1
    T1 = idx.array("T")
    p = check.recursion_properties(T0, T1, res["anchors"], nsep, 20)
    assert p["all"], p
    # the anchor set and the lower-cased text are the CPU recursion's (reveal.c:731-1338 with the benchmark callbacks), bit for bit
    g = check.compare_with_golden(golden(name), anchors=res["anchors"], T_final=T1)
    assert g["all"], g
    # 1 % substitutions: about one anchor per substitution-free stretch, nearly everything anchored
    assert p["anchors"] > L // 150
    if G == 2:
        assert p["anchored_bp"] > 0.9 * 2 * L / 2
    with pytest.raises(TypeError):
        idx.SA                                                # main SA/LCP are gone after align (reveal.c:1279-1284)


@pytest.mark.parametrize("cfg", ["C4", "C3"])
def test_cascade_and_level_pipeline_agree_at_full_size(monkeypatch, cfg):
    """C4 / C3: the anchor cascade (rv_cascade.hip for two samples, rv_cascade_multi.hip for ten; what an untraced run takes) and the
    level pipeline it stands in for (RV_NO_CASCADE: scan / split / bubble_sort of every level, reveal.c:731-1338 step by step) give
    the same anchors, counters and final text at full size"""
    from reveal_amd import reveallib
    L, G, seed = CONFIGS[cfg]
    seqs = synth.genomes(L, G, seed=seed)
    idx = reveallib.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    out = []
    for off in (False, True):
        if off:
            idx.set_option("RV_NO_CASCADE", 1)      # (a switch of this handle: the library does not read the environment)
        idx.construct()
        res = idx.align_builtin(20, 2)
        info = idx.cascade_info()
        assert info["done"] == (not off), info
        l, o, pos = res["anchors"]
        assert (np.diff(o) == G).all()                        # every anchor has a member in every sample
        first, second = pos[0::G], pos[1::G]
        order = np.lexsort((second, first))
        members = pos.reshape(-1, G)[order]
        out.append((np.asarray(l)[order], members, idx.array("T").copy(),
                    {k: res["stats"][k] for k in (("steps", "splits", "anchored_bp") if G == 2 else ("splits", "anchored_bp"))}))
    for which, o in zip(("cascade", "level pipeline"), out):      # each path against the CPU recursion's digests
        ll, mm, tt, _ = o
        g = check.compare_with_golden(golden(cfg), anchors=(ll, np.arange(0, len(mm.ravel()) + 1, G), mm.ravel()), T_final=tt)
        assert g["all"], (which, g)
    a, b = out
    assert a[3] == b[3], (a[3], b[3])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])


def test_c4_with_indels_equals_the_cpu_digests():
    """2 x 250 Mbp with the reference simulator's mutation model (20 % of the events indels, reveal_amd/synth.py after utils/simulate.py:17-77):
    the second sample leaves the fixed diagonal within a few hundred bases, construct follows piecewise diagonals from seeds (rv_construct.hip
    k_diag_bits_tab) -- SA, LCP, the anchor set and the final text equal the CPU path's at full size (tests/golden/fullsize.json C4_indel_seed42)"""
    from reveal_amd import reveallib
    rec = check.golden_record(250_000_000, 2, 42, 0.2)
    assert rec is not None
    seqs = synth.genomes(250_000_000, 2, seed=42, indelfrac=0.2)
    T0 = np.frombuffer(b"$".join(seqs) + b"$", dtype=np.uint8)
    assert len(T0) == rec["n"] and check.array_digest(T0) == rec["sha_input"]
    idx = reveallib.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    idx.construct()
    st = idx.sa_stats()
    assert st["diag_table"] == 1 and st["sorted_elems"] < 0.7 * idx.n, st          # the table was used, most twins left the sort
    g = check.compare_with_golden(rec, SA=idx.array("SA"), LCP=idx.array("LCP"))
    assert g["all"], g
    res = idx.align_builtin(20, 2)
    g = check.compare_with_golden(rec, anchors=res["anchors"], T_final=idx.array("T"))
    assert g["all"], (g, idx.cascade_info())
