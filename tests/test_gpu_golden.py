"""The HIP path against the golden vectors captured from the reference's own C
(tests/golden/vectors.json, oracle/gen_golden.py) -- no oracle involved."""
import hashlib

import pytest

from helpers import csr_tuples, feed, golden, golden_inputs, sha_arr, sha_json, trace_digests

pytestmark = pytest.mark.gpu
SETS = golden()


@pytest.mark.parametrize("label", sorted(SETS))
def test_hip_matches_reference_vectors(label):
    from reveal_amd import reveallib, reveallib64
    g = SETS[label]
    inputs = golden_inputs(g)
    idx = feed((reveallib64 if g["sa64"] else reveallib).index(), inputs)
    assert idx.n == g["n"] and idx.nsep == g["nsep"] and sorted(list(x) for x in idx.nodes) == g["nodes"]
    idx.construct()
    assert hashlib.sha256(idx.T.encode("latin-1")).hexdigest() == g["sha_T"]
    assert sha_arr(idx.array("SA")) == g["sha_SA"]
    assert sha_arr(idx.array("LCP")) == g["sha_LCP"]
    mums = [[m[0], list(m[1]), m[2]] for m in idx.getmums(g["getmums"]["minl"])]
    assert len(mums) == g["getmums"]["count"] and sha_json(mums) == g["getmums"]["sha"]
    if "getmultimums" in g:
        mm = idx.getmultimums(minlength=g["minl"], minn=g["minn"])
        assert len(mm) == g["getmultimums"]["count"] and sha_json(mm) == g["getmultimums"]["sha"]
    got = idx.align_builtin(g["minl"], g["minn"], trace=True)
    st, sa_, na, bp = trace_digests(got["trace"])
    rg = g["recursion"]
    assert len(got["trace"]) == rg["steps"] and na == rg["anchors"] and bp == rg["anchored_bp"]
    assert sa_ == rg["sha_anchors"]
    assert st == rg["sha_trace"]           # every sub-index: intervals, size, scan result, choice, SA and LCP arrays
    assert hashlib.sha256(idx.T.encode("latin-1")).hexdigest() == rg["sha_finalT"]
    assert got["stats"]["maxdepth"] == rg["maxdepth"]


@pytest.mark.parametrize("label", sorted(k for k in SETS if "extract" in SETS[k]))
def test_hip_extract_matches_reference_vectors(label):
    """extract() against the output of the reference's own extract() (reveal.c:1386-1505); the reference never
    writes the new SA[0], so it is compared from rank 1 on"""
    from reveal_amd import reveallib, reveallib64
    g = SETS[label]
    idx = feed((reveallib64 if g["sa64"] else reveallib).index(), golden_inputs(g))
    idx.construct()
    e = g["extract"]
    idx.extract([tuple(x) for x in e["intervals"]])
    assert idx.n == e["n"]
    assert sha_arr(idx.array("SA")[1:]) == e["sha_SA1"] and sha_arr(idx.array("LCP")) == e["sha_LCP"]
    assert hashlib.sha256(idx.T.encode("latin-1")).hexdigest() == e["sha_T"]
