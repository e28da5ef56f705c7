"""The host side of the anchor picker against the reference's own Python functions (reveal/schemes.py chain / trim_overlap /
segment, reveal/utils.py gapcost): tests/golden/chain_vectors.json holds their outputs (oracle/gen_chain_golden.py executes
them, converted in memory, in the build container).  rv_chain is plain host C++ inside the HIP library: no GPU needed."""
import json
import os

import pytest

from helpers import GOLD
from reveal_amd import schemes


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(GOLD, "chain_vectors.json")) as f:
        return json.load(f)


def test_gapcost(vec):
    for c in vec["gapcost"]:
        for model in ("sumofpairs", "star-avg", "star-med"):
            assert schemes.gapcost(c["a"], c["b"], model) == c[model]


def test_chain_matches_the_reference_decision_for_decision(vec):
    assert len(vec["chain"]) >= 90
    for c in vec["chain"]:
        keys = c["keys"]
        mums = [(r[0], r[1], dict(zip(keys, r[2:]))) for r in c["mums"]]
        left, right = (0, 0, dict(zip(keys, c["left"]))), (0, 0, dict(zip(keys, c["right"])))
        got = schemes.chain(list(mums), left, right, wscore=c["wscore"], wpen=c["wpen"])
        # the reference returns the chain right to left (schemes.py:97-105); its caller reverses it
        want = [((r[0], r[1], dict(zip(keys, r[2:-1]))), r[-1]) for r in c["path"]][::-1]
        assert got == want


def test_chain_models_and_errors():
    mums = [(10, 2, {0: 5, 1: 7}), (8, 2, {0: 30, 1: 29}), (12, 2, {0: 60, 1: 66})]
    left, right = (0, 0, {0: -1, 1: -1}), (0, 0, {0: 100, 1: 100})
    for model in ("sumofpairs", "star-avg", "star-med"):
        got = schemes.chain(list(mums), left, right, gcmodel=model)
        assert [m for m, _ in got] == mums                      # collinear: everything chains
    assert schemes.chain([], left, right) == []
    with pytest.raises(RuntimeError):                           # a match in front of `left` has no predecessor
        schemes.chain([(5, 2, {0: -4, 1: 3})], left, right)


def _idx(rows):
    return [(l, n, tuple((s, p) for s, p in spd)) for l, n, spd in rows]


def test_trim_overlap(vec):
    for c in vec["trim_overlap"]:
        assert schemes.trim_overlap(_idx(c["mums"])) == _idx(c["out"])


def test_segment(vec):
    for c in vec["segment"]:
        assert schemes.segment(_idx(c["mums"])) == _idx(c["out"])
