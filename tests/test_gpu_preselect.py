"""Anchor pre-selection inside the library (`index.preselect(maxmums)` / rv_set_preselect).

The reference's picker starts every call the same way (schemes.py:227, 240, 245-247, 287-289): keep the matches
with n == idx.nsamples, sort by length, cap at `maxmums`.  `head()` below restates those lines literally; the tests
check that (1) the list the library hands out with pre-selection on IS head(full list) as a list, order included,
(2) a picker that begins with head() therefore sees the same thing and the alignment is the same -- anchors,
callback trace, final text -- with and without it."""
import pytest

from helpers import assemble, fa, feed, oracle, synth
from reveal_amd import rem

pytestmark = pytest.mark.gpu


def head(mums, nsamples, maxmums):
    """schemes.py:227 / :240 / :245-247 / :287-289, the same two stable sorts and the tail slice"""
    mmums = [m for m in mums if m[1] == nsamples]
    mmums.sort(key=lambda m: m[0], reverse=True)
    rel = list(mmums)
    rel.sort(key=lambda m: (m[1], m[0]))
    if len(rel) > maxmums:
        rel = rel[-maxmums:]
    return rel


def capped_picker(maxmums, log):
    """a picker whose choice depends on the whole capped list (not only on its longest member): of the capped
    matches the one whose smallest coordinate is the median"""
    def pick(mums, idx, precomputed=False, minlength=0):
        rel = head(mums, idx.nsamples, maxmums)
        log.append((idx.depth, min(idx.nodes), len(mums), tuple(rel)))
        if not rel:
            return ()
        rel = sorted(rel, key=lambda m: (min(p for _, p in m[2]), m[0]))
        return (rel[len(rel) // 2], [], [])
    return pick


def run(inputs, maxmums, preselect, minl, sa64=False):
    from reveal_amd import reveallib, reveallib64
    idx = feed((reveallib64 if sa64 else reveallib).index(), inputs)
    idx.construct()
    if preselect:
        idx.preselect(maxmums)
    log, anchors = [], []

    def galign(i, mum):
        r = rem.linear_graphalign(i, mum)
        if r is not None:
            anchors.append(mum)
        return r
    idx.align(capped_picker(maxmums, log), galign, minl=minl, minn=2)
    return idx.T, sorted(anchors), sorted(log, key=lambda r: (r[0], r[1])), log


class View(object):       # what the callbacks read from an index
    def __init__(self, nodes, nsamples, depth):
        self.nodes, self.nsamples, self.depth = set(nodes), nsamples, depth


def oracle_recursion(inputs, maxmums, minl, sa64=False):
    """the same callbacks driving the CPU oracle alone (scan, splitindex = label + split + lower-casing + bubble_sort)"""
    T, nsep, nodes = assemble(inputs)
    ns = len(inputs)
    O = oracle(sa64)
    c = O.construct(T, nsep, ns)
    log, anchors = [], []
    pick = capped_picker(maxmums, log)
    queue = [(c["SA"], c["LCP"], sorted(nodes), ns, 0)]
    while queue:
        sa, lcp, nd, nsub, depth = queue.pop()
        if ns > 2:
            l, n, off, so, pos = O.getmultimums(c["tbuf"], sa, lcp, c["SO"], c["nsep"], ns, minl, 2)
            mums = [(int(l[k]), int(n[k]), tuple((int(so[q]), int(pos[q])) for q in range(off[k], off[k + 1]))) for k in range(len(l))]
        else:
            l, a, b = O.getmums(c["tbuf"], sa, lcp, c["nsep"], minl, rem=True, nT=len(c["SA"]))
            mums = [(int(l[k]), 2, ((0, int(a[k])), (1, int(b[k])))) for k in range(len(l))]
        view = View(nd, nsub, depth)
        r = pick(mums, view)
        if r == ():
            continue
        ga = rem.linear_graphalign(view, r[0])
        if ga is None:
            continue
        anchors.append(r[0])
        lead, trail, match, rest = ga[:4]
        kids = O.splitindex(c["tbuf"], sa, lcp, c["SAi"], c["SO"], c["nsep"], ns, lead, trail, match, rest)
        for k, ivs in enumerate((lead, trail, rest)):
            if kids[k] is not None and len(kids[k][0]) > 1:
                queue.append((kids[k][0], kids[k][1], ivs, kids[k][2], depth + 1))
    return bytes(c["tbuf"][:len(T)]).decode("latin-1"), sorted(anchors), sorted(log, key=lambda r: (r[0], r[1]))


CASES = [
    ("pair", fa("1a", "1b"), 20, False),
    ("pair64", fa("1a", "1b"), 20, True),
    ("three", fa("1a", "1b", "1c"), 20, False),
    ("five", fa("1a", "1b", "1c", "1d", "1e"), 20, False),
    ("synthetic3", [g.decode() for g in synth.genomes(60000, 3)], 12, False),
]


@pytest.mark.parametrize("maxmums", [1, 3, 50])
@pytest.mark.parametrize("name,inputs,minl,sa64", CASES)
def test_same_alignment_with_preselection(name, inputs, minl, sa64, maxmums):
    T0, an0, tr0, _ = run(inputs, maxmums, False, minl, sa64)
    T1, an1, tr1, _ = run(inputs, maxmums, True, minl, sa64)
    assert T0 == T1
    assert an0 == an1 and len(an0) > 0
    To, ano, tro = oracle_recursion(inputs, maxmums, minl, sa64)
    assert T1 == To and an1 == ano
    tr0, tr1, tro = ([r for r in t if r[2] > 0] for t in (tr0, tr1, tro))      # (calls with an empty list: nothing to compare)
    assert [(d, b, rel) for d, b, _, rel in tr1] == [(d, b, rel) for d, b, _, rel in tro]
    # per callback: same capped list (order included); only the length of the list that crossed into Python differs
    assert [(d, b, rel) for d, b, _, rel in tr0] == [(d, b, rel) for d, b, _, rel in tr1]
    assert sum(r[2] for r in tr1) <= sum(r[2] for r in tr0)
    if maxmums == 1 and name != "pair64":
        assert sum(r[2] for r in tr1) < sum(r[2] for r in tr0)


@pytest.mark.parametrize("host_filter", [False, True])
@pytest.mark.parametrize("name,inputs,minl,sa64", CASES[:4])
def test_handed_out_list_is_the_reference_head(monkeypatch, name, inputs, minl, sa64, host_filter):
    """the list itself: with pre-selection the picker receives exactly head(full list) -- in emission order, which
    the stable sorts of head() turn into the same sorted list -- or the whole list where nothing spans every sample.
    host_filter: RV_PRESEL_HOST=1, every match copied to the host and filtered there (the scan kernel drops the matches
    that are not in every sample of their sub-index otherwise, and scans the sub-indices without such a match again)"""
    if host_filter:
        monkeypatch.setenv("RV_PRESEL_HOST", "1")
    maxmums = 4
    full, got = {}, {}

    def recorder(store):
        def pick(mums, idx, precomputed=False, minlength=0):
            store[(idx.depth, min(idx.nodes))] = (list(mums), idx.nsamples)
            return rem.bench_mumpicker(mums, idx)
        return pick
    from reveal_amd import reveallib, reveallib64
    for store, on in ((full, False), (got, True)):
        idx = feed((reveallib64 if sa64 else reveallib).index(), inputs)
        idx.construct()
        if on:
            idx.preselect(maxmums)
        idx.align(recorder(store), rem.linear_graphalign, minl=minl, minn=2)
    assert full.keys() == got.keys() and len(full) > 3
    capped = 0
    for key, (mums, ns) in full.items():
        sel = got[key][0]
        want = head(mums, ns, maxmums)
        if not any(m[1] == ns for m in mums):
            assert sel == mums                       # schemes.py:229-232 segments over all of them
            continue
        assert head(sel, ns, maxmums) == want
        assert sorted(sel) == sorted(want) and len(sel) <= maxmums
        pos = [mums.index(m) for m in sel]
        assert pos == sorted(pos)                    # emission order kept
        capped += len(sel) < len(mums)
    assert capped > 0


def test_preselect_off_again_and_errors():
    from reveal_amd import reveallib
    idx = feed(reveallib.index(), fa("1a", "1b"))
    idx.construct()
    idx.preselect(2)
    idx.preselect(0)
    seen = []

    def pick(mums, i, precomputed=False, minlength=0):
        seen.append(len(mums))
        return rem.bench_mumpicker(mums, i)
    idx.align(pick, rem.linear_graphalign, minl=20, minn=2)
    assert max(seen) > 2
    with pytest.raises(reveallib.error):
        feed(reveallib.index(), fa("1a", "1b")).preselect(-1)
    # getmums of a constructed index is never pre-selected
    idx = feed(reveallib.index(), fa("1a", "1b"))
    idx.construct()
    n = len(idx.getmums(20))
    idx.preselect(1)
    assert len(idx.getmums(20)) == n and n > 1


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("maxmums", [1, 4, 50])
@pytest.mark.parametrize("name,inputs,minl", [("1a1b", fa("1a", "1b"), 20), ("synthetic2", [g.decode() for g in synth.genomes(80000, 2, seed=3)], 12)])
def test_cap_applied_on_the_device(monkeypatch, name, inputs, minl, maxmums, sa64):
    """two samples: the `maxmums` longest matches of every sub-index are chosen on the device (pair_topk, rv_api.hip) once a level holds
    RV_PRESEL_DEV_MIN records -- 0 here: every level -- instead of on the host from everything that was copied (10^9: never on the device).
    The picker receives the same lists either way, order included: of equal lengths the later emitted stay (minl 12 on random text: many
    ties); they are head(list) of themselves and no longer than the cap; the alignment is the same.  (What the host's cap hands out is
    compared with the reference's own selection in the tests above and in tools/fuzz_preselect.py, which runs both forms.)"""
    host, dev = {}, {}

    def recorder(store):
        def pick(mums, idx, precomputed=False, minlength=0):
            store[(idx.depth, min(idx.nodes))] = (list(mums), idx.nsamples)
            return rem.bench_mumpicker(mums, idx)
        return pick
    from reveal_amd import reveallib, reveallib64
    texts = []
    for store, dev_min in ((host, "1000000000"), (dev, "0")):
        monkeypatch.setenv("RV_PRESEL_DEV_MIN", dev_min)
        idx = feed((reveallib64 if sa64 else reveallib).index(), inputs)
        idx.construct()
        idx.preselect(maxmums)
        idx.align(recorder(store), rem.linear_graphalign, minl=minl, minn=2)
        texts.append(idx.T)
    assert texts[0] == texts[1]
    assert host.keys() == dev.keys() and len(host) > 3
    full_lists = 0
    for key, (mums, ns) in host.items():
        sel = dev[key][0]
        assert sel == mums, key
        assert len(sel) <= maxmums and head(sel, ns, maxmums) == sorted(sel, key=lambda m: (m[1], m[0]))
        full_lists += len(sel) == maxmums
    assert full_lists > 0
