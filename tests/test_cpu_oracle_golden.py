"""CPU suite: the oracle (oracle/reveal_oracle.c) against the golden vectors that
oracle/gen_golden.py captured from the reference's own C (oracle/_ref), plus the
known-answer vectors of SURVEY.md 8(c).  No GPU, no /root/reference needed."""
import numpy as np
import pytest

import os

from helpers import GOLD, ROOT, assemble, csr_tuples, golden, golden_inputs, oracle, sha_arr, sha_json, trace_digests

SETS = golden()


@pytest.mark.parametrize("label", sorted(SETS))
def test_oracle_matches_reference_vectors(label):
    g = SETS[label]
    inputs = golden_inputs(g)
    O = oracle(g["sa64"])
    T, nsep, nodes = assemble(inputs)
    assert len(T) == g["n"] and nsep == g["nsep"] and [list(x) for x in nodes] == g["nodes"]
    c = O.construct(T, nsep, len(inputs))
    assert sha_arr(c["SA"]) == g["sha_SA"]
    assert sha_arr(c["LCP"]) == g["sha_LCP"]
    assert int(c["LCP"].max()) == g["maxlcp"]
    if "SA" in g:
        assert [int(x) for x in c["SA"]] == g["SA"] and [int(x) for x in c["LCP"]] == g["LCP"]
    l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, g["getmums"]["minl"])
    mums = [[int(l[k]), [int(a[k]), int(b[k])], 0] for k in range(len(l))]
    assert len(mums) == g["getmums"]["count"] and sha_json(mums) == g["getmums"]["sha"]
    if "getmultimums" in g:
        mm = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, len(inputs), g["minl"], g["minn"]))
        assert len(mm) == g["getmultimums"]["count"] and sha_json(mm) == g["getmultimums"]["sha"]
    r = O.align_bench(c, nodes, g["minl"], g["minn"], trace_cap=g["recursion"]["steps"] + 8)
    st, sa_, na, bp = trace_digests(r["trace"])
    rg = g["recursion"]
    assert len(r["trace"]) == rg["steps"] and na == rg["anchors"] and bp == rg["anchored_bp"]
    assert sa_ == rg["sha_anchors"] and st == rg["sha_trace"]
    import hashlib
    assert hashlib.sha256(r["T"]).hexdigest() == rg["sha_finalT"]
    assert r["stats"]["maxdepth"] == rg["maxdepth"]


@pytest.mark.parametrize("label", sorted(k for k in SETS if "extract" in SETS[k]))
def test_oracle_extract_matches_reference_vectors(label):
    """ro_extract against the reference's own extract() (reveal.c:1386-1505) on the longest full match"""
    import hashlib
    g = SETS[label]
    inputs = golden_inputs(g)
    O = oracle(g["sa64"])
    T, nsep, nodes = assemble(inputs)
    c = O.construct(T, nsep, len(inputs))
    e = g["extract"]
    sa, lcp, iv = O.extract(c["tbuf"], c["SA"], c["LCP"], c["SAi"], nsep, [tuple(x) for x in e["intervals"]], nT=len(T))
    assert len(sa) == e["n"] and sha_arr(sa[1:]) == e["sha_SA1"] and sha_arr(lcp) == e["sha_LCP"]
    assert sa[0] == c["SA"][0]
    assert hashlib.sha256(bytes(c["tbuf"][:len(T)])).hexdigest() == e["sha_T"]


def test_oracle_splitindex_is_one_step_of_the_recursion():
    """ro_splitindex (reveal.c:1515-1748) from interval lists = label + split + bubble_sort of the aligner: driving it
    with the bench callbacks reproduces the anchors of ro_align (which the golden trace pins)"""
    from reveal_amd import rem
    g = SETS["1a1b"]
    O = oracle(False)
    T, nsep, nodes = assemble(golden_inputs(g))
    c = O.construct(T, nsep, 2)
    ref = O.align_bench(O.construct(T, nsep, 2), nodes, 20, 2, trace_cap=g["recursion"]["steps"] + 8)

    class View(object):
        pass
    queue, anchors = [(c["SA"], c["LCP"], sorted(nodes), 2)], []
    while queue:
        sa, lcp, nd, ns = queue.pop()
        l, a, b = O.getmums(c["tbuf"], sa, lcp, nsep, 20, rem=True, nT=len(T))
        v = View(); v.nodes, v.nsamples = nd, ns
        r = rem.bench_mumpicker([(int(l[k]), 2, ((0, int(a[k])), (1, int(b[k])))) for k in range(len(l))], v)
        if r == ():
            continue
        lead, trail, match, rest = rem.linear_graphalign(v, r[0])[:4]
        anchors.append((r[0][0], min(p for _, p in r[0][2]), 2))
        kids = O.splitindex(c["tbuf"], sa, lcp, c["SAi"], None, nsep, 2, lead, trail, match, rest)
        for k, ivs in zip(kids, (lead, trail, rest)):
            if k is not None:
                queue.append((k[0], k[1], ivs, k[2]))
    assert sha_json(sorted(anchors)) == g["recursion"]["sha_anchors"]
    assert bytes(c["tbuf"][:len(T)]) == ref["T"]


@pytest.mark.parametrize("label", ["known2", "t1t2", "1a1b", "d1d2", "1a1b_64"])
def test_own_suffix_sorter_matches_reference_sa(label):
    """the oracle's own prefix-doubling sorter (used when oracle/_ref is absent) gives divsufsort's SA"""
    g = SETS[label]
    O = oracle(g["sa64"])
    T, nsep, nodes = assemble(golden_inputs(g))
    tb = O.textbuf(T)
    SA = O.suffix_array(tb, own=True)
    assert sha_arr(SA) == g["sha_SA"] and O.sufcheck(tb, SA) == 0


def test_known_answer_vectors():
    """captured from the reference on reveal/tests/test_reveal.py:37's input"""
    O = oracle(False)
    T, nsep, nodes = assemble(["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"])
    assert T == b"ACTTGCTAGCTAGTCAG$ACTAGCTAGCTAGTGAG$" and nsep == [17] and nodes == [(0, 17), (18, 35)]
    c = O.construct(T, nsep, 2)
    assert list(c["SA"]) == [35, 17, 18, 0, 33, 15, 21, 7, 25, 11, 29, 14, 19, 5, 23, 9, 27, 1, 34, 16, 32, 4, 22, 8, 26, 12, 30, 20, 6, 24, 10, 28, 13, 31, 3, 2]
    assert list(c["LCP"]) == [0, 0, 0, 3, 1, 2, 2, 6, 7, 2, 3, 0, 1, 8, 9, 4, 5, 2, 0, 1, 1, 1, 10, 5, 6, 1, 2, 0, 7, 8, 3, 4, 1, 1, 2, 1]
    l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 1)
    assert list(zip(l, a, b)) == [(3, 0, 18), (10, 4, 22), (2, 3, 31)]
    T, nsep, nodes = assemble(["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"])
    c = O.construct(T, nsep, 3)
    assert len(T) == 54 and nsep == [17, 35]
    mm = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, 3, 2, 2))
    assert mm == [(9, 2, ((0, 0), (2, 36))), (3, 3, ((1, 18), (0, 0), (2, 36))), (10, 2, ((0, 4), (1, 22))),
                  (7, 2, ((2, 46), (0, 10))), (4, 3, ((2, 46), (0, 10), (1, 28))), (2, 3, ((1, 31), (0, 3), (2, 39)))]
    me = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, 3, 4, 3, mems=True))
    assert me == [(5, 3, ((0, 4), (1, 22), (2, 40), (0, 8), (1, 26))), (4, 3, ((2, 46), (0, 10), (1, 28)))]


def test_lcp_closed_form():
    """LCP[k] = min(plain lcp with the predecessor, distance to the first '$'/'N') -- the form the HIP kernel uses"""
    O = oracle(False)
    T, nsep, nodes = assemble([SETS["d1d2"]["inputs"][0], SETS["d1d2"]["inputs"][1]] if False else golden_inputs(SETS["d1d2"]))
    c = O.construct(T, nsep, 2)
    SA, LCP = c["SA"], c["LCP"]
    t = np.frombuffer(T, dtype=np.uint8)
    stop = (t == ord("$")) | (t == ord("N"))
    nxt = np.full(len(t) + 1, len(t), dtype=np.int64)
    for i in range(len(t) - 1, -1, -1):
        nxt[i] = i if stop[i] else nxt[i + 1]
    rng = np.random.default_rng(1)
    for k in rng.integers(1, len(SA), size=3000):
        a, b = int(SA[k - 1]), int(SA[k])
        h = 0
        while a + h < len(t) and b + h < len(t) and t[a + h] == t[b + h]:
            h += 1
        assert LCP[k] == min(h, nxt[b] - b)


def test_split_skips_min_update_for_unlabelled_ranks():
    """reveal.c:616-620: a rank with D==0 `continue`s past the running-minimum update"""
    O = oracle(False)
    SA = np.array([10, 11, 12, 13, 14], dtype=np.int32)
    LCP = np.array([0, 5, 1, 7, 6], dtype=np.int32)
    D = np.array([1, 1, 0, 1, 1], dtype=np.uint8)       # rank 2 unlabelled
    SAi = np.zeros(32, dtype=np.int32)
    kids = O.split(SA, LCP, D, SAi, 4, 0, 0)
    assert list(kids[0][0]) == [10, 11, 13, 14]
    # LCP[3]=7 is never folded in (rank 2 skipped its update), LCP[2]=1 is
    assert list(kids[0][1]) == [0, 5, 1, 6]


def _bubble_closed_form(SA, LCP, B):
    """the restatement rv_bubble.hip implements (see its header): chains of l', movers, landing sites,
    final arrangement = for every non-mover k: [movers landing at k by t ascending], k"""
    n = len(SA)
    lp = list(LCP)
    mover = [False] * n
    for i in range(n):
        if SA[i] < B and SA[i] + lp[i] > B:
            if i == 0:
                if n > 1:
                    lp[1] = B - SA[0]
            else:
                mover[i] = True
                if i + 1 < n and lp[i] < lp[i + 1]:
                    lp[i + 1] = lp[i]
        elif i + 1 < n and SA[i] < B and SA[i] + lp[i + 1] > B and lp[i + 1] > lp[i]:
            lp[i + 1] = B - SA[i]
    groups = {}
    for j in range(n):
        if not mover[j]:
            continue
        t, k = B - SA[j], j - 1
        while k > 0 and (mover[k] or lp[k] >= t):
            k -= 1
        groups.setdefault(k, []).append(j)
    outS, outL = [], []
    for r in range(n):
        if mover[r]:
            continue
        gap = lp[r]
        for j in sorted(groups.get(r, []), key=lambda j: B - SA[j]):
            outS.append(SA[j]); outL.append(gap); gap = B - SA[j]
        outS.append(SA[r]); outL.append(gap)
    return outS, outL


def test_bubble_closed_form():
    """bubble_sort's sequential loop (reveal.c:666-727, oracle ro_bubble_sort) has a closed form; the GPU's
    data-parallel rounds implement that form, so it is pinned here against the oracle on arbitrary arrays"""
    import random
    import numpy as np
    O = oracle(False)
    rng = random.Random(7)
    for it in range(4000):
        n = rng.randint(1, 40)
        maxl = rng.choice([2, 5, 12, 30, 100])
        SA = rng.sample(range(3 * n), n)
        LCP = [rng.randint(0, maxl) for _ in range(n)]
        if it % 5:
            LCP[0] = 0
        cuts = [rng.randint(0, 3 * n) for _ in range(rng.randint(1, 3))]
        sa = np.array(SA, dtype=np.int32); lcp = np.array(LCP, dtype=np.int32)
        sai = np.zeros(3 * n + 1, dtype=np.int32)
        sai[sa] = np.arange(n, dtype=np.int32)
        O.bubble_sort(sa, lcp, sai, cuts)
        cs, cl = list(SA), list(LCP)
        for B in cuts:
            cs, cl = _bubble_closed_form(cs, cl, B)
        assert cs == sa.tolist() and cl == lcp.tolist(), (SA, LCP, cuts)
        assert (sai[sa] == np.arange(n)).all()


def test_reference_aligner_itself_pins_the_oracle(tmp_path):
    """The reference's own aligner() (reveal.c:731-1338, through index.align, interface.c:293-415) executed for real with
    the benchmark callbacks as Python 3 functions: its per-callback trace, anchors and final text equal ro_align's and the
    digests in vectors.json.  Needs oracle/_ref/reveallib.so (`make -C oracle refmod`: the reference's sources built as the
    CPython module they define); where that is absent (no /root/reference at build time) the test is skipped."""
    import gzip
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "oracle"))
    import pin_oracle as P
    if P.load_refmod(False) is None:
        pytest.skip("oracle/_ref/reveallib.so not built (make -C oracle refmod needs /root/reference)")
    gold = golden()

    def files(names):
        out = []
        for x in names:
            src = os.path.join(here, "golden", x + ".fa.gz")
            if not os.path.exists(src):
                out.append(x)
                continue
            dst = tmp_path / (x + ".fa")
            with gzip.open(src, "rt") as f:
                dst.write_text(f.read())
            out.append(str(dst))
        return out
    P.check.failed = 0
    for label, sa64 in (("known2", False), ("t1t2", False), ("1a1b", False), ("1a1b_64", True), ("1a1b1c", False), ("5way", False), ("d1d2", False)):
        g = gold[label]
        P.pin_aligner(label, files(g["inputs"]), sa64, minl=g["minl"], minn=g["minn"], golden=g)
    assert P.check.failed == 0


def test_fullsize_golden_file_is_reproducible():
    """tests/golden/fullsize.json: the record of the smallest configuration comes out of oracle/gen_fullsize_golden.py again, digest
    for digest (the large ones take the same code minutes to hours; their inputs are pinned by sha_input)"""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_fullsize_golden as g
    from oracle import ref_ctypes
    if not ref_ctypes.available(False):
        pytest.skip("oracle/_ref is not built here")
    want = json.load(open(os.path.join(GOLD, "fullsize.json")))["C2_seed42"]
    got = g.run("C2_seed42")
    for k in want:
        if k != "cpu_seconds":
            assert got[k] == want[k], k
