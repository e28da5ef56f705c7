"""The recursion (scan -> pick -> label -> split -> bubble_sort) on the GPU against the
CPU oracle: every sub-index visited must have the same intervals, size, scan result,
chosen match and child SA/LCP arrays (reveallib/reveal.c:731-1338).  The GPU visits
level by level, the oracle LIFO; records are matched by (depth, smallest interval begin)."""
import random

import numpy as np
import pytest

from helpers import assemble, csr_tuples, fa, feed, oracle, synth
from reveal_amd import rem

pytestmark = pytest.mark.gpu

FIELDS = ("key", "n", "depth", "nsamples", "nnodes", "nmums", "picked", "l", "mn", "sp_min", "h_sa", "h_lcp", "h_mums")


def mod(sa64):
    from reveal_amd import reveallib, reveallib64
    return reveallib64 if sa64 else reveallib


def oracle_run(inputs, minl, minn, sa64=False):
    T, nsep, nodes = assemble(inputs)
    O = oracle(sa64)
    c = O.construct(T, nsep, len(inputs))
    return O.align_bench(c, nodes, minl, minn, trace_cap=4 * len(T) // max(minl, 1) + 1000), T


def compare(inputs, minl=20, minn=2, sa64=False):
    ref, T = oracle_run(inputs, minl, minn, sa64)
    idx = feed(mod(sa64).index(), inputs)
    idx.construct()
    got = idx.align_builtin(minl, minn, trace=True)
    rt, gt = ref["trace"], got["trace"]
    assert len(rt) == len(gt), (len(rt), len(gt))
    ro = np.lexsort((rt["key"], rt["depth"])); go = np.lexsort((gt["key"], gt["depth"]))
    for f in FIELDS:
        a, b = rt[f][ro].astype(np.uint64), gt[f][go].astype(np.uint64)
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, "field %s differs at %d of %d sub-indices; first: depth=%d key=%d ref=%d got=%d" % (
            f, len(bad), len(a), rt["depth"][ro][bad[0]], rt["key"][ro][bad[0]], a[bad[0]], b[bad[0]])
    # anchors as a set
    rl, rn, roff, rpos = ref["anchors"]
    ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    gl, goff, gpos = got["anchors"]
    ga = sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl)))
    assert ra == ga
    assert idx.T.encode("latin-1") == ref["T"]            # lower-case mask
    assert got["stats"]["splits"] == ref["stats"]["nsplits"]
    assert got["stats"]["steps"] == ref["stats"]["nsteps"]
    return idx, got, ref


@pytest.mark.parametrize("name,inputs,minl", [
    ("known", ["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"], 2),
    ("t1t2", fa("t1", "t2"), 1),
    ("d1d2", fa("d1", "d2"), 20),
    ("1e1b", fa("1e", "1b"), 20),
    ("1a1a", fa("1a", "1a"), 20),
    ("1a1b", fa("1a", "1b"), 20),
    ("1a1b_m10", fa("1a", "1b"), 10),
])
def test_pairwise_recursion(name, inputs, minl):
    compare(inputs, minl)


def test_pairwise_recursion_64bit():
    compare(fa("1a", "1b"), 20, sa64=True)


@pytest.mark.parametrize("L", [2000, 200000, 1000000])
def test_synthetic_pair(L):
    seqs = [g.decode() for g in synth.genomes(L, 2)]
    idx, got, ref = compare(seqs, 20)
    # test15's invariant (reveal/tests/test_reveal.py:150-159) in array form: every anchored
    # range is lower case, everything else still upper case, and the text spells the input
    assert idx.T.upper().encode("latin-1") == assemble(seqs)[0]


@pytest.mark.parametrize("name,inputs", [
    ("known3", ["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"]),
    ("1a1b1c", fa("1a", "1b", "1c")),
    ("5way", fa("1a", "1b", "1c", "1d", "1e")),
])
def test_multi_recursion(name, inputs):
    compare(inputs, 20 if name != "known3" else 2, 2)


@pytest.mark.parametrize("inputs,minl,minn", [
    (fa("1a", "1b", "1c"), 20, 2), (fa("1a", "1b", "1c"), 2, 2), (fa("1a", "1b", "1c"), 0, 2), (fa("1a", "1b", "1c"), 20, 3),
    (fa("1a", "1b", "1c", "1d", "1e"), 20, 2), (fa("1a", "1b", "1c", "1d", "1e"), 2, 2), (fa("1a", "1b", "1c", "1d", "1e"), 20, 5),
    (["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"], 2, 2),
])
def test_getmultimums(inputs, minl, minn):
    T, nsep, nodes = assemble(inputs)
    O = oracle(False)
    c = O.construct(T, nsep, len(inputs))
    ref = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, len(inputs), minl, minn))
    idx = feed(mod(False).index(), inputs)
    idx.construct()
    assert idx.getmultimums(minlength=minl, minn=minn) == ref


@pytest.mark.parametrize("inputs,minl,minn,sa64", [
    (fa("1a", "1b", "1c"), 20, 2, False), (fa("1a", "1b", "1c"), 8, 2, False), (fa("1a", "1b", "1c"), 20, 3, False), (fa("1a", "1b", "1c"), 4, 3, False),
    (fa("1a", "1b", "1c", "1d", "1e"), 20, 2, False), (fa("1a", "1b", "1c", "1d", "1e"), 12, 5, False), (fa("1a", "1b", "1c", "1d", "1e"), 6, 4, True),
    (["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"], 2, 2, False),
    (["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"], 1, 3, False),
    (fa("1a", "1a", "1b"), 15, 3, False),          # a repeated sample: intervals with many members from few samples (the `continue` path)
])
def test_getmultimems(inputs, minl, minn, sa64):
    """reveal.c:292-434 incl. its order dependence (a multi-MEM covering fewer than minn samples leaves the loop body before
    `lb = i_lb`, reveal.c:340-342 / :362): same list, same order as the oracle's restatement (pinned against the reference)"""
    T, nsep, nodes = assemble(inputs)
    O = oracle(sa64)
    c = O.construct(T, nsep, len(inputs))
    ref = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, len(inputs), minl, minn, mems=True))
    idx = feed(mod(sa64).index(), inputs)
    idx.construct()
    got = idx.getmultimems(minlength=minl, minn=minn)
    assert len(got) == len(ref)
    assert got == ref


def test_getmultimems_synthetic_and_two_samples():
    seqs = [g.decode() for g in synth.genomes(120000, 4, seed=11)]
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, 4)
    idx = feed(mod(False).index(), seqs)
    idx.construct()
    for minl, minn in ((18, 2), (14, 4), (25, 3)):
        ref = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, 4, minl, minn, mems=True))
        assert idx.getmultimems(minlength=minl, minn=minn) == ref and len(ref) > 100
    # two samples: every qualifying interval counts one "sample" and is skipped for minn >= 2 (reveal.c:268, :340): empty
    pair = feed(mod(False).index(), seqs[:2])
    pair.construct()
    assert pair.getmultimems(minlength=20, minn=2) == []


def test_getmultimems_runs_of_every_kind():
    """rv_mems.hip round 5: a stack machine per run of LCP values of minl and more -- a thread for the short ones, a wavefront for those of
    more than 2048 ranks or deeper than a thread's stack.  Random families of 3-6 samples with what makes runs long, deep and private to
    one sample (the `continue` of reveal.c:340-342): homopolymers, tandem arrays, copies inside one sample, an N run, a repeated sample;
    small minl (long runs everywhere) and minl = 1 (the whole index is a handful of runs)"""
    rng = random.Random(99)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    cases = 0
    for case in range(14):
        k = rng.choice([3, 3, 4, 5, 6])
        L = rng.choice([1500, 6000, 20000])
        base = rnd(L)
        seqs = []
        for s in range(k):
            v = list(base)
            for p in range(len(v)):
                if rng.random() < 0.01:
                    v[p] = rng.choice("ACGT")
            v = "".join(v)
            kind = rng.randrange(6)
            at = rng.randrange(len(v))
            if kind == 0:
                v = v[:at] + rng.choice("ACGT") * rng.choice([40, 700, 5000]) + v[at:]            # a homopolymer: one run, as deep as it is long
            elif kind == 1:
                v = v[:at] + rnd(rng.randint(2, 30)) * rng.choice([5, 60, 400]) + v[at:]           # a tandem array
            elif kind == 2:
                u = rnd(rng.choice([30, 300]))
                for _ in range(rng.choice([2, 6, 40])):                                            # copies private to this sample
                    q = rng.randrange(len(v)); v = v[:q] + u + v[q:]
            elif kind == 3:
                v = v[:at] + "N" * rng.choice([3, 50, 3000]) + v[at:]
            seqs.append(v)
        if rng.random() < 0.3:
            seqs[-1] = seqs[0]
        T, nsep, nodes = assemble(seqs)
        sa64 = case % 5 == 4
        O = oracle(sa64)
        c = O.construct(T, nsep, k)
        idx = feed(mod(sa64).index(), seqs)
        idx.construct()
        for minl, minn in ((20, 2), (rng.choice([3, 6, 12]), rng.choice([2, 3])), (1, 2), (30, k)):
            ref = csr_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, k, minl, minn, mems=True))
            got = idx.getmultimems(minlength=minl, minn=minn)
            assert len(got) == len(ref), (case, k, minl, minn)
            assert got == ref, (case, k, minl, minn)
            cases += 1
    assert cases == 56


@pytest.mark.parametrize("par_min", [0, 64, 1000])
@pytest.mark.parametrize("name,inputs,minl,sa64", [
    ("t1t2", fa("t1", "t2"), 1, False),
    ("d1d2", fa("d1", "d2"), 20, False),
    ("1e1b", fa("1e", "1b"), 20, False),
    ("1a1a", fa("1a", "1a"), 20, False),       # identical inputs: windows beyond RV_PB_CAP -> sequential kernels inside a parallel round
    ("1a1b_m10", fa("1a", "1b"), 10, False),
    ("1a1b_64", fa("1a", "1b"), 20, True),
    ("1a1b1c", fa("1a", "1b", "1c"), 20, False),
    ("5way", fa("1a", "1b", "1c", "1d", "1e"), 20, False),
])
def test_parallel_bubble_rounds(monkeypatch, par_min, name, inputs, minl, sa64):
    """the data-parallel bubble_sort (rv_bubble.hip, closed form of reveal.c:666-727) forced onto
    small leading children too: same child SA/LCP as the oracle's literal loop, sub-index by sub-index"""
    monkeypatch.setenv("RV_BUBBLE_PAR_MIN", str(par_min))
    monkeypatch.setenv("RV_NO_LEAF", "1")
    compare(inputs, minl, 2, sa64=sa64)


@pytest.mark.parametrize("name,inputs,minl", [
    ("1a1b", fa("1a", "1b"), 20),
    ("1a1b_m10", fa("1a", "1b"), 10),
    ("d1d2", fa("d1", "d2"), 20),
    ("synth_pair", None, 20),
    ("1a1b1c", fa("1a", "1b", "1c"), 20),
    ("5way", fa("1a", "1b", "1c", "1d", "1e"), 20),
    ("synth12", (30000, 12), 20),       # picker windows from LDS (up to 16 samples)
    ("synth20", (20000, 20), 20),       # more than 16 samples: the picker's general form
    ("synth70", (4000, 70), 15),        # more than 64 samples: sample sets no longer fit a bit mask
])
@pytest.mark.parametrize("cascade", [False, True])
def test_untraced_run_same_anchors(monkeypatch, name, inputs, minl, cascade):
    """align_builtin(trace=False) is what bench.py times: leaf kernel, device-side picker (pair) and pre-selection (multi)
    are on, the host never sees the full MUM lists -- the anchor set and the final text must not change.
    cascade False: the level pipeline for two-sample runs as well (RV_NO_CASCADE); True: the default, rv_cascade.hip first"""
    if not cascade:
        monkeypatch.setenv("RV_NO_CASCADE", "1")
    if inputs is None:
        inputs = [g.decode() for g in synth.genomes(400000, 2)]
    elif isinstance(inputs, tuple):
        inputs = [g.decode() for g in synth.genomes(inputs[0], inputs[1], seed=3)]
    ref, T = oracle_run(inputs, minl, 2)
    idx = feed(mod(False).index(), inputs)
    idx.construct()
    got = idx.align_builtin(minl, 2, trace=False)
    rl, rn, roff, rpos = ref["anchors"]
    ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    gl, goff, gpos = got["anchors"]
    ga = sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl)))
    assert ra == ga
    assert idx.T.encode("latin-1") == ref["T"]
    assert got["stats"]["splits"] == ref["stats"]["nsplits"]
    if len(inputs) == 2:      # two samples: every sub-index the reference visits is counted (the leaf kernel counts the ones it does not scan)
        assert got["stats"]["steps"] == ref["stats"]["nsteps"]


@pytest.mark.parametrize("acap", [0, 1, 3])
@pytest.mark.parametrize("name,inputs,minl", [("1a1b", fa("1a", "1b"), 20), ("1a1b_m10", fa("1a", "1b"), 10), ("d1d2", fa("d1", "d2"), 20), ("synth_pair", None, 20)])
def test_leaf_anchor_staging_overflow(monkeypatch, acap, name, inputs, minl):
    """RV_LEAF_ACAP: the leaf kernel stages that many anchors per workgroup in LDS (256 by default) and writes the rest straight
    to the output, one reservation each -- forced here with a staging area of 0 / 1 / 3 anchors"""
    monkeypatch.setenv("RV_LEAF_ACAP", str(acap))
    monkeypatch.setenv("RV_NO_CASCADE", "1")      # (the level pipeline's leaf launches; the cascade's own launch: tests/test_gpu_cascade.py, tools/fuzz.py)
    if inputs is None:
        inputs = [g.decode() for g in synth.genomes(400000, 2)]
    ref, T = oracle_run(inputs, minl, 2)
    idx = feed(mod(False).index(), inputs)
    idx.construct()
    got = idx.align_builtin(minl, 2, trace=False)
    rl, rn, roff, rpos = ref["anchors"]
    ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    gl, goff, gpos = got["anchors"]
    ga = sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl)))
    assert ra == ga
    assert idx.T.encode("latin-1") == ref["T"]
    assert got["stats"]["splits"] == ref["stats"]["nsplits"] and got["stats"]["steps"] == ref["stats"]["nsteps"]


@pytest.mark.parametrize("name,inputs,minl", [("1a1b_m10", fa("1a", "1b"), 10), ("1a1a", fa("1a", "1a"), 20), ("5way", fa("1a", "1b", "1c", "1d", "1e"), 20)])
def test_bubble_rounds_two_pass_form(monkeypatch, name, inputs, minl):
    """RV_PB_TWO_PASS=1: copy-out + scatter through the scratch arrays instead of the one-pass shift (k_pb_shift) -- same arrays"""
    monkeypatch.setenv("RV_BUBBLE_PAR_MIN", "64")
    monkeypatch.setenv("RV_NO_LEAF", "1")
    monkeypatch.setenv("RV_PB_TWO_PASS", "1")
    compare(inputs, minl, 2)


def test_chunked_carry_scan(monkeypatch):
    """levels above 64 M ranks scan their tile summaries in chunks (reduce / scan / apply): force that path on a small input"""
    monkeypatch.setenv("RV_CARRY_CH", "3")
    compare([g.decode() for g in synth.genomes(150000, 2)], 20)
    compare(fa("1a", "1b", "1c"), 20, 2)


@pytest.mark.parametrize("name,inputs,minl,sa64", [
    ("1a1b", fa("1a", "1b"), 20, False),
    ("1a1a", fa("1a", "1a"), 20, False),
    ("d1d2", fa("d1", "d2"), 20, False),
    ("1a1b_64", fa("1a", "1b"), 20, True),
    ("5way", fa("1a", "1b", "1c", "1d", "1e"), 20, False),
    ("synth3", None, 20, False),
])
def test_lds_resident_bubble(monkeypatch, name, inputs, minl, sa64):
    """children of at most 8192 ranks bubbled on LDS copies of their arrays, on a second stream (the path levels with
    thousands of small children take): forced on for every level"""
    monkeypatch.setenv("RV_BUBBLE_LDS_ALWAYS", "1")
    monkeypatch.setenv("RV_NO_LEAF", "1")
    if inputs is None:
        inputs = [g.decode() for g in synth.genomes(200000, 3)]
    compare(inputs, minl, 2, sa64=sa64)


def test_sequential_bubble_kept(monkeypatch):
    """the one-workgroup-per-child kernels stay the fallback (and the small-level path): keep them covered in multi mode"""
    monkeypatch.setenv("RV_BUBBLE_NO_JOIN", "1")
    monkeypatch.setenv("RV_BUBBLE_PAR_MIN", str(1 << 40))
    compare(fa("1a", "1b", "1c"), 20, 2)
    compare([g.decode() for g in synth.genomes(200000, 2)], 20)


def test_parallel_bubble_synthetic(monkeypatch):
    monkeypatch.setenv("RV_BUBBLE_PAR_MIN", "0")
    seqs = [g.decode() for g in synth.genomes(300000, 3)]
    compare(seqs, 20)


def test_python_callbacks_same_anchors():
    """index.align() with Python callbacks of the reference's signatures (reveal.c:839-999)
    gives the same anchors and final text as the native driver"""
    inputs = fa("1a", "1b")
    ref, T = oracle_run(inputs, 20, 2)
    idx, anchors = rem.align_genomes(inputs, minlength=20, minn=2)
    rl, rn, roff, rpos = ref["anchors"]
    ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    ga = sorted((int(m[0]), tuple(sorted(int(p) for _, p in m[2]))) for m in anchors)
    assert ra == ga
    assert idx.T.encode("latin-1") == ref["T"]
    with pytest.raises(TypeError):
        idx.SA            # main SA/LCP are gone after align (reveal.c:1279-1284)


@pytest.mark.parametrize("env", ["RV_NO_EARLY_SPLIT", "RV_NO_EARLY_BUBBLE"])
@pytest.mark.parametrize("name,inputs", [("1a1b1c", fa("1a", "1b", "1c")), ("synth12", (30000, 12)), ("synth5_big", (400000, 5))])
def test_untraced_multi_host_paths(monkeypatch, name, inputs, env):
    """more than two samples, untraced: the level's split (and, without round-sized children, lower-casing and bubble) are queued from
    decisions taken on the device (rv_decide.hip k_decide_multi) while the host rebuilds them; with the switches the host-built
    tables drive the same kernels -- same anchors, same text, equal to the oracle's"""
    monkeypatch.setenv(env, "1")
    if isinstance(inputs, tuple):
        inputs = [g.decode() for g in synth.genomes(inputs[0], inputs[1], seed=3)]
    ref, T = oracle_run(inputs, 20, 2)
    idx = feed(mod(False).index(), inputs)
    idx.construct()
    got = idx.align_builtin(20, 2, trace=False)
    rl, rn, roff, rpos = ref["anchors"]
    ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    gl, goff, gpos = got["anchors"]
    ga = sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl)))
    assert ra == ga
    assert idx.T.encode("latin-1") == ref["T"]
