"""Every configuration of tests/golden/fullsize.json that a test can afford twice (both recursion paths): the HIP path's SA, LCP,
anchor set and final text against the digests the CPU path left there (oracle/gen_fullsize_golden.py: the reference's divsufsort,
compute_lcp and the recursion of reveal.c:731-1338 restated, run in the build container).  C4 / C3 at their full size are in
tests/test_gpu_fullsize.py."""
import json
import os

import numpy as np
import pytest

from helpers import synth, GOLD
from reveal_amd import check

pytestmark = pytest.mark.gpu

RECORDS = json.load(open(os.path.join(GOLD, "fullsize.json")))
NAMES = [k for k, r in RECORDS.items() if r["n"] <= 110_000_000]


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("cascade", [True, False])
@pytest.mark.parametrize("name", NAMES)
def test_digests(name, cascade, sa64):
    """(the 64-bit library's arrays are narrowed to the digests' 32-bit layout: every position is below 2^31.  Every record through both libraries and
    both recursion paths -- the multi-sample and level-pipeline combinations of reveallib64 were skipped up to round 5)"""
    from reveal_amd import reveallib, reveallib64
    r = RECORDS[name]
    reveallib = reveallib64 if sa64 else reveallib
    seqs = synth.family(r["L"], r["genomes"], seed=r["seed"], snp=r["snp"], indelfrac=r["indelfrac"], repeats=r.get("repeats", 0.0), nruns=r.get("nruns", 0))
    T0 = np.frombuffer(b"$".join(seqs) + b"$", dtype=np.uint8)
    assert len(T0) == r["n"] and check.array_digest(T0) == r["sha_input"]        # the generator still makes the bytes the CPU saw
    idx = reveallib.index()
    if not cascade:
        idx.set_option("RV_NO_CASCADE", 1)
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    idx.construct()
    rec = dict(r, name=name)
    g = check.compare_with_golden(rec, SA=idx.array("SA").astype(np.int32), LCP=idx.array("LCP").astype(np.int32))
    assert g["all"], g
    assert idx.maxlcp == r["maxlcp"]
    res = idx.align_builtin(r["minl"], r["minn"])
    info = idx.cascade_info()
    g = check.compare_with_golden(rec, anchors=res["anchors"], T_final=idx.array("T"))
    assert g["all"], (g, info)
    assert res["stats"]["splits"] == r["anchors"] and res["stats"]["anchored_bp"] == r["anchored_bp"]
