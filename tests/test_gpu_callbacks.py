"""index.align() with Python callbacks: seeded children (skipmums) and the error paths of aligner()
(reveallib/reveal.c:783-792, 802-837, 859-868, 976-985; interface.c:293-415).

The seeded runs are compared with digests of the REFERENCE's own aligner() driven by the very same callback objects
(tests/golden/vectors.json "seeded", made by oracle/gen_golden.py from the reference built as the CPython module it defines):
every mumpicker call -- sub-index, its SA / LCP, the list it was handed, whether it was precomputed, the pick -- the anchors
and the final text."""
import hashlib

import pytest

from helpers import callback_trace_digest, fa, feed, golden_inputs, golden_seeded, traced_callbacks
from reveal_amd import rem

pytestmark = pytest.mark.gpu


def mod(sa64):
    from reveal_amd import reveallib, reveallib64
    return reveallib64 if sa64 else reveallib


@pytest.mark.parametrize("label", ["1a1b", "1a1b1c", "5way", "1a1b_64"])
def test_seeded_children_match_the_reference_aligner(label):
    g = golden_seeded()[label]
    idx = feed(mod(g["sa64"]).index(), golden_inputs(g))
    idx.construct()
    pick, galign, trace = traced_callbacks(rem.seeding_mumpicker, rem.linear_graphalign)
    idx.align(pick, galign, threads=0, minl=g["minl"], minn=g["minn"])
    d = callback_trace_digest(trace)
    assert d["calls"] == g["calls"] and d["precomputed_calls"] == g["precomputed_calls"] > 0 and d["anchors"] == g["anchors"]
    assert d["sha_anchors"] == g["sha_anchors"]
    assert d["sha_trace"] == g["sha_trace"]
    assert hashlib.sha256(idx.T.encode("latin-1")).hexdigest() == g["sha_finalT"]


def test_seeded_children_are_not_scanned():
    """a child with skipmums gets exactly that list (same object), precomputed=True; its own scan result is never built"""
    idx = feed(mod(False).index(), fa("1a", "1b"))
    idx.construct()
    seen = []

    def pick(mums, sub, precomputed=False, minlength=0):
        seen.append((precomputed, mums is sub.skipmums if precomputed else len(sub.skipmums) == 0))
        return rem.seeding_mumpicker(mums, sub, precomputed=precomputed, minlength=minlength)
    idx.align(pick, rem.linear_graphalign, minl=20, minn=2)
    assert any(p for p, _ in seen) and all(ok for _, ok in seen)


def _fresh():
    idx = feed(mod(False).index(), fa("1a", "1b"))
    idx.construct()
    return idx


def _still_usable(idx):
    """after a failed align() the recursion state is closed (rv_align_end): construct + align work again"""
    idx.construct()
    res = idx.align_builtin(20, 2)
    assert res["stats"]["splits"] == 548


def test_mumpicker_not_callable():
    idx = _fresh()
    with pytest.raises(TypeError, match="mumpicker isn't callable"):          # reveal.c:783-792
        idx.align(None, rem.linear_graphalign, minl=20, minn=2)
    _still_usable(idx)


def test_mumpicker_returns_no_tuple():
    idx = _fresh()
    with pytest.raises(TypeError, match="call to mumpicker failed"):          # reveal.c:859-868
        idx.align(lambda mums, sub, precomputed=False, minlength=0: [1, 2, 3], rem.linear_graphalign, minl=20, minn=2)
    _still_usable(idx)


def test_graphalign_returns_no_tuple():
    idx = _fresh()
    with pytest.raises(TypeError, match="call to graphalign failed"):         # reveal.c:976-985
        idx.align(rem.bench_mumpicker, lambda sub, mum: "nonsense", minl=20, minn=2)
    _still_usable(idx)


def test_graphalign_none_ends_the_branch():
    """graphalign returning None: the sub-index ends there without an error (reveal.c:960-974)"""
    idx = _fresh()
    calls = []

    def galign(sub, mum):
        calls.append(sub.depth)
        return None if sub.depth >= 2 else rem.linear_graphalign(sub, mum)
    idx.align(rem.bench_mumpicker, galign, minl=20, minn=2)
    assert max(calls) == 2 and calls.count(0) == 1


def test_raising_callbacks_propagate_and_close_the_run():
    class Boom(Exception):
        pass

    def bad_pick(mums, sub, precomputed=False, minlength=0):
        if sub.depth == 3:
            raise Boom("picker")
        return rem.bench_mumpicker(mums, sub, precomputed=precomputed, minlength=minlength)

    def bad_align(sub, mum):
        if sub.depth == 2:
            raise Boom("graphalign")
        return rem.linear_graphalign(sub, mum)
    idx = _fresh()
    with pytest.raises(Boom, match="picker"):
        idx.align(bad_pick, rem.linear_graphalign, minl=20, minn=2)
    _still_usable(idx)
    idx.construct()
    with pytest.raises(Boom, match="graphalign"):
        idx.align(rem.bench_mumpicker, bad_align, minl=20, minn=2)
    _still_usable(idx)
    # overlapping intervals from graphalign are refused by the library, not silently mislabelled
    idx.construct()
    with pytest.raises(mod(False).error):
        idx.align(rem.bench_mumpicker, lambda sub, mum: ([(0, 50)], [(40, 90)], [(100, 120)], [], None, None, None), minl=20, minn=2)
    _still_usable(idx)
