"""The anchor cascade (reveal_amd/csrc/rv_cascade.hip) against the CPU oracle's literal recursion (reveal.c:731-1338 with the
benchmark callbacks): untraced two-sample runs decided from the top-level match list must give the reference's anchors,
final text and counters -- on the reference's fixtures, on synthetic genomes, and on random inputs with repeats, N runs and
indels, with the cascade on, off (RV_NO_CASCADE) and forced to give up."""
import os
import random
import sys

import numpy as np
import pytest

from helpers import ROOT, assemble, fa, feed, oracle, synth

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def mod(sa64):
    from reveal_amd import reveallib, reveallib64
    return reveallib64 if sa64 else reveallib


def aset(a):
    if len(a) == 4:
        l, n, off, pos = a
    else:
        l, off, pos = a
    return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))


def oracle_run(inputs, minl, sa64=False):
    T, nsep, nodes = assemble(inputs)
    O = oracle(sa64)
    c = O.construct(T, nsep, len(inputs))
    return O.align_bench(c, nodes, minl, 2)


def check(inputs, minl, sa64=False, want_done=None):
    ref = oracle_run(inputs, minl, sa64)
    idx = feed(mod(sa64).index(), inputs)
    idx.construct()
    got = idx.align_builtin(minl, 2)
    info = idx.cascade_info()
    assert aset(got["anchors"]) == aset(ref["anchors"])
    assert idx.T.encode("latin-1") == ref["T"]
    assert got["stats"]["splits"] == ref["stats"]["nsplits"] and got["stats"]["steps"] == ref["stats"]["nsteps"]
    assert got["stats"]["anchored_bp"] == ref["stats"]["anchored_bp"]
    if want_done is not None:
        assert info["done"] == want_done, info
    return info


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("name,inputs,minl", [("1a1b", fa("1a", "1b"), 20), ("1a1b_m10", fa("1a", "1b"), 10), ("1e1b", fa("1e", "1b"), 20),
                                              ("1a1a", fa("1a", "1a"), 20), ("synth", (60000, 2), 20), ("synth_big", (1500000, 2), 20)])
def test_cascade_equals_the_literal_recursion(name, inputs, minl, sa64):
    if isinstance(inputs, tuple):
        inputs = [g.decode() for g in synth.genomes(inputs[0], inputs[1], seed=11)]
    info = check(inputs, minl, sa64)
    if name.startswith("synth"):      # unrelated repeats of random text are short: the cascade decides the whole run
        assert info["done"] and info["levels"] > 3 and info["matches"] > 100, info


def test_cascade_off_and_on_agree(monkeypatch):
    inputs = [g.decode() for g in synth.genomes(300000, 2, seed=5)]
    on = check(inputs, 20, want_done=True)
    monkeypatch.setenv("RV_NO_CASCADE", "1")
    off = check(inputs, 20, want_done=False)
    assert on["subindices"] > 0 and off["levels"] == 0


def test_cascade_gives_up_cleanly(monkeypatch):
    """tandem arrays longer than the leaf kernel's size with different point mutations in the two samples: the sub-index around
    them cannot be decided from the match list (its best match is no longer than its repeats) and is too large to be rebuilt --
    the attempt must leave nothing behind, and the level pipeline's result is the reference's (the attempts that would decide it
    from its witnesses or rebuild it for the level pipeline switched off)"""
    monkeypatch.setenv("RV_CASCADE_DANGER", "0")
    rng = random.Random(3)
    gave_up = 0
    for case in range(4):
        base = "".join(rng.choice("ACGT") for _ in range(30000))
        unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([7, 23, 61])))
        arr = unit * (6000 // len(unit))

        def mutated(s, every):
            s = list(s)
            for p in range(rng.randint(0, every), len(s), every):
                s[p] = rng.choice("ACGT")
            return "".join(s)
        a = base[:15000] + mutated(arr, 97) + base[15000:]
        b = base[:15000] + mutated(arr, 89) + base[15000:]
        info = check([a, b], 20)
        gave_up += (not info["done"]) and info["matches"] > 0
    assert gave_up > 0


def test_cascade_with_undecided_subindices():
    """repeats of minl characters and more inside the gaps between anchors: sub-indices the match list cannot decide are rebuilt
    from their text (k_cas_rank / k_cas_emit) and finished by the leaf kernel"""
    rng = random.Random(8)
    seen = 0
    for case in range(6):
        L = rng.choice([30000, 120000])
        base = [rng.choice("ACGT") for _ in range(L)]
        rep = "".join(rng.choice("ACGT") for _ in range(rng.choice([25, 60, 150])))
        for _ in range(rng.randint(5, 40)):      # copies of one repeat, some of them in diverged surroundings
            p = rng.randint(0, L - 200)
            base[p:p + len(rep)] = list(rep)
        base = "".join(base)
        other = list(base)
        for _ in range(L // 30):
            p = rng.randint(0, L - 1)
            other[p] = rng.choice("ACGT")
        if case % 2:
            other[L // 3:L // 3] = list("N" * rng.randint(1, 50))
        info = check([base, "".join(other)], rng.choice([15, 20, 30]))
        seen += info["undecided"] if info["done"] else 0
    assert seen > 0


def test_cascade_random_inputs():
    """the generator of tools/fuzz.py (SNPs, indels, tandem repeats, N runs, identical copies), two samples"""
    from fuzz import make_case
    rng = random.Random(2026)
    done = 0
    for _ in range(40):
        seqs, minl = make_case(rng)
        seqs = [s for s in seqs[:2]]
        if min(len(s) for s in seqs) == 0:
            continue
        done += check(seqs, minl)["done"]
    assert done >= 10


# ---- more than two samples (rv_cascade_multi.hip) ----------------------------------------------------------------------
@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("name,inputs,minl", [("1a1b1c", fa("1a", "1b", "1c"), 20), ("synth3", (200000, 3), 20), ("synth5", (400000, 5), 20),
                                              ("synth10", (150000, 10), 20), ("synth16", (30000, 16), 15), ("synth17", (20000, 17), 15)])
def test_multi_cascade_equals_the_literal_recursion(name, inputs, minl, sa64):
    if isinstance(inputs, tuple):
        inputs = [g.decode() for g in synth.genomes(inputs[0], inputs[1], seed=13)]
    ref = oracle_run(inputs, minl, sa64)
    idx = feed(mod(sa64).index(), inputs)
    idx.construct()
    got = idx.align_builtin(minl, 2)
    info = idx.cascade_info()
    assert aset(got["anchors"]) == aset(ref["anchors"])
    assert idx.T.encode("latin-1") == ref["T"]
    assert got["stats"]["splits"] == ref["stats"]["nsplits"] and got["stats"]["anchored_bp"] == ref["stats"]["anchored_bp"]
    if name in ("synth3", "synth5", "synth10"):               # one sequence per sample, unrelated repeats are short: decided by the cascade
        assert info["done"] and info["matches"] > 10, info
    if name == "synth17":                                     # more samples than the cascade takes
        assert not info["done"]


def test_multi_cascade_random_inputs():
    """the generator of tools/fuzz.py with three and four samples: indels make sub-indices lack samples (undecided, rebuilt from the
    text and finished by the level pipeline), repeats make them undecided or make the cascade give up"""
    from fuzz import make_case
    rng = random.Random(77)
    done = und = 0
    n = 0
    while n < 40:
        seqs, minl = make_case(rng)
        if len(seqs) < 3 or min(len(s) for s in seqs) == 0:
            continue
        n += 1
        ref = oracle_run(seqs, minl)
        idx = feed(mod(False).index(), seqs)
        idx.construct()
        got = idx.align_builtin(minl, 2)
        info = idx.cascade_info()
        assert aset(got["anchors"]) == aset(ref["anchors"]), (n, info)
        assert idx.T.encode("latin-1") == ref["T"]
        assert got["stats"]["splits"] == ref["stats"]["nsplits"]
        done += info["done"]; und += info["undecided"] if info["done"] else 0
    assert done >= 10 and und > 0


def _multi_equal(seqs, minl, sa64=False):
    ref = oracle_run(seqs, minl, sa64)
    idx = feed(mod(sa64).index(), seqs)
    idx.construct()
    got = idx.align_builtin(minl, 2)
    info = idx.cascade_info()
    assert aset(got["anchors"]) == aset(ref["anchors"]), info
    assert idx.T.encode("latin-1") == ref["T"]
    assert got["stats"]["splits"] == ref["stats"]["nsplits"] and got["stats"]["anchored_bp"] == ref["stats"]["anchored_bp"]
    return info


@pytest.mark.parametrize("sa64", [False, True])
def test_multi_cascade_rebuilds_large_undecided_subindices(monkeypatch, sa64):
    """a sample that lost kilobases leaves the others' kilobases behind as one sub-index that lacks a sample: above the 8192 ranks a workgroup rebuilds
    in LDS they go through global memory (k_casmb_keys / place / emit) instead of ending the cascade; with what makes suffixes tie there -- an N run,
    a tandem array, a copy of another stretch, ends that agree (all of them shorter than the matches of the five samples: a repeat longer than those
    leaves the ROOT undecided)"""
    rng = random.Random(5)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    base = rnd(260000)
    unit = rnd(37)
    base = base[:50000] + "N" * 40 + base[50040:52000] + unit * 6 + base[52000 + 37 * 6:]         # inside what sample 1 loses
    base = base[:120000] + base[124000:124150] + base[120150:]                                      # a copy of a stretch, inside what sample 2 loses
    seqs = [base]
    for s in range(1, 5):
        seqs.append(_snp(rng, base, 0.01))
    seqs[1] = seqs[1][:48000] + seqs[1][55000:]                  # 7 kb gone: 4 x 7 kb left with the others
    seqs[2] = seqs[2][:118000] + seqs[2][131000:]                # 13 kb
    seqs[3] = seqs[3][:200000] + rnd(6000) + seqs[3][200000:]    # an insertion: a sub-index with one sample
    seqs[4] = seqs[4][:230000] + seqs[4][233000:233120] + seqs[4][230000:]      # a short duplication
    info = _multi_equal(seqs, 20, sa64)
    assert info["done"] and info["rebuilt_ranks"] > 4 * 13000, info
    monkeypatch.setenv("RV_CASM_NO_BIG", "1")      # up to round 4: the cascade gave up and the level pipeline ran from the top
    info = _multi_equal(seqs, 20, sa64)
    assert not info["done"] and "above the size" in info["why"], info
    monkeypatch.delenv("RV_CASM_NO_BIG")
    monkeypatch.setenv("RV_CASM_BIG_MIN", "0")     # every undecided sub-index through global memory
    info = _multi_equal(seqs, 20, sa64)
    assert info["done"]
    monkeypatch.delenv("RV_CASM_BIG_MIN")
    # low-complexity text in such a sub-index -- thousands of suffixes that agree for kilobases -- is not compared pair by pair: the cascade notices,
    # takes back what it lower-cased and leaves the run to the level pipeline
    # (an N run: the LCP array stops at N, so it is no repeat for the cascade's bounds -- a tandem array of this length would leave the ROOT undecided --, but its
    #  suffixes share their bytes for as long as the run lasts)
    seqs = [s[:52000] + "N" * 2500 + s[52000:] if k != 1 else s for k, s in enumerate(seqs)]
    info = _multi_equal(seqs, 20, sa64)
    assert not info["done"] and "sharing their first bytes" in info["why"], info


def test_multi_cascade_random_inputs_through_global_memory(monkeypatch):
    """test_multi_cascade_random_inputs with every undecided sub-index, however small, rebuilt by the kernels for the large ones"""
    monkeypatch.setenv("RV_CASM_BIG_MIN", "0")
    from fuzz import make_case
    rng = random.Random(78)
    done = und = n = 0
    while n < 40:
        seqs, minl = make_case(rng)
        if len(seqs) < 3 or min(len(s) for s in seqs) == 0:
            continue
        n += 1
        info = _multi_equal(seqs, minl, sa64=(n % 4 == 0))
        done += info["done"]; und += info["undecided"] if info["done"] else 0
    assert done >= 10 and und > 0


# ---- two samples through the interval cascade (the second attempt of rv_align.hip builtin_cascade) ---------------------------------
@pytest.mark.parametrize("name,inputs,minl", [("1a1b", fa("1a", "1b"), 20), ("1a1b_m10", fa("1a", "1b"), 10), ("synth", (300000, 2), 20)])
def test_pairs_through_the_interval_cascade(monkeypatch, name, inputs, minl):
    """RV_CASCADE_SECOND=2: two samples straight through rv_cascade_multi.hip (bound = repeats inside one sample; undecided
    sub-indices to the level pipeline) -- the literal recursion's result"""
    monkeypatch.setenv("RV_CASCADE_SECOND", "2")
    if isinstance(inputs, tuple):
        inputs = [g.decode() for g in synth.genomes(inputs[0], inputs[1], seed=3)]
    info = check(inputs, minl)
    if name == "synth":
        assert info["done"]


def test_second_attempt_takes_what_the_leaf_kernel_cannot(monkeypatch):
    """tandem arrays of a few thousand bases with different point mutations: the sub-index around them is undecided and larger than the
    leaf kernel takes (the first attempt gives up), but small enough to be rebuilt for the level pipeline (the interval cascade; the
    attempt that decides it from its witnesses switched off)"""
    monkeypatch.setenv("RV_CASCADE_DANGER", "0")
    rng = random.Random(12)
    second = 0
    for case in range(5):
        base = "".join(rng.choice("ACGT") for _ in range(40000))
        unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([11, 23, 47])))
        arr = unit * (rng.choice([1500, 2200, 3000]) // len(unit))

        def mutated(x, every):
            x = list(x)
            for p in range(rng.randint(0, every), len(x), every):
                x[p] = rng.choice("ACGT")
            return "".join(x)
        a = base[:20000] + mutated(arr, 97) + base[20000:]
        b = base[:20000] + mutated(arr, 89) + base[20000:]
        info = check([a, b], 20)
        second += info["done"] and info["rebuilt_ranks"] > 2048
    assert second > 0


# ---- large undecided sub-indices decided from their repeat witnesses (k_cas_dwalk) ---------------------------------------------------
def _snp(rng, s, rate):
    s = list(s)
    for p in range(len(s)):
        if rng.random() < rate:
            s[p] = rng.choice("ACGT")
    return "".join(s)


def _with_copies(rng, base, unit, copies):
    out, at = [], 0
    for p in sorted(rng.sample(range(len(base)), copies)):
        out.append(base[at:p]); out.append(unit); at = p
    out.append(base[at:])
    return "".join(out)


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("kind", ["element", "operon", "array", "array_both", "low_complexity"])
def test_repeats_as_long_as_the_matches(kind, sa64):
    """interspersed repeats as long as the longest matches (copies of a mobile element, of an operon), tandem arrays: the sub-indices at
    the top are not decided by the match list and far too large to rebuild -- the second attempt decides them from the witnesses"""
    rng = random.Random({"element": 1, "operon": 2, "array": 3, "array_both": 4, "low_complexity": 5}[kind])
    base = "".join(rng.choice("ACGT") for _ in range(150000))
    if kind == "element":
        a = _with_copies(rng, base, "".join(rng.choice("ACGT") for _ in range(1200)), 12)
        b = _snp(rng, a, 0.01)
    elif kind == "operon":
        a = _with_copies(rng, base, "".join(rng.choice("ACGT") for _ in range(4000)), 5)
        b = _snp(rng, a, 0.01)
    elif kind == "array":
        unit = "".join(rng.choice("ACGT") for _ in range(37))
        a = base[:50000] + _snp(rng, unit * 80, 0.02) + base[50000:]
        b = _snp(rng, base[:50000] + unit * 80 + base[50000:], 0.01)
    elif kind == "array_both":
        unit = "".join(rng.choice("ACGT") for _ in range(61))
        a = base[:50000] + _snp(rng, unit * 70, 0.01) + base[50000:]
        b = _snp(rng, base[:50000] + _snp(rng, unit * 64, 0.01) + base[50000:], 0.01)
    else:
        a = base[:70000] + "A" * 700 + "AC" * 900 + base[70000:]
        b = _snp(rng, base[:70000] + "A" * 650 + "AC" * 1000 + base[70000:], 0.01)
    info = check([a, b], 20, sa64)
    if kind in ("element", "operon"):
        assert info["done"] and info["decided_from_witnesses"] > 0, info


def test_witness_decisions_on_random_inputs(monkeypatch):
    """RV_CASCADE_DANGER=2: the first attempt already decides every undecided sub-index from its witnesses where it can"""
    from fuzz import make_case
    monkeypatch.setenv("RV_CASCADE_DANGER", "2")
    rng = random.Random(909)
    solved = 0
    for _ in range(40):
        seqs, minl = make_case(rng)
        seqs = [s for s in seqs[:2]]
        if min(len(s) for s in seqs) == 0:
            continue
        solved += check(seqs, minl)["decided_from_witnesses"]
    assert solved > 0


def _write_fasta(path, contigs):
    with open(path, "w") as f:
        for k, s in enumerate(contigs):
            f.write(">c%d\n%s\n" % (k, s))
    return str(path)


@pytest.mark.parametrize("sa64", [False, True])
def test_cascade_with_several_sequences_per_sample(tmp_path, monkeypatch, sa64):
    """draft assemblies: a sample is a FASTA file of several contigs (one '$' each, utils.py:325-350).  The sub-indices with more than one
    interval per sample form the chain root -> rest -> rest ..., decided on the host from the match list; their leading / trailing
    children are the roots of the device cascade (rv_cascade.hip "the lineage of rest sub-indices").  Anchors, text and counters equal
    the oracle's literal recursion: contigs cut at different places in the two samples, in another order, contigs without a partner,
    contigs shorter than minl, one sample in one piece; and inputs whose left-over contigs share a repeat or only chance matches (the chain stops
    at that member, which becomes the level pipeline's frontier: one split of the root with the consumed sequences dropped)"""
    rng = random.Random(17)

    def rnd(L):
        return "".join(rng.choice("ACGT") for _ in range(L))

    def snp(s, rate):
        a = list(s)
        for p in range(len(a)):
            if rng.random() < rate:
                a[p] = rng.choice("ACGT")
        return "".join(a)

    def cut(s, k):
        at = sorted(rng.sample(range(200, len(s) - 200), k - 1))
        return [s[i:j] for i, j in zip([0] + at, at + [len(s)])]
    base = rnd(240000)
    var = snp(base, 0.01)
    rep = rnd(90)
    cases = []
    c1, c2 = cut(base, 5), cut(var, 6)
    rng.shuffle(c2)
    cases.append(("cut differently, shuffled", c1, c2, True))
    cases.append(("one piece against seven", [base], cut(var, 7), True))
    cases.append(("unrelated extra contigs", cut(base, 3) + [rnd(5000)], [rnd(7000)] + cut(var, 4), True))
    cases.append(("tiny contigs", cut(base, 3) + ["ACGTACGT", "A"], ["ACG"] + cut(var, 3) + ["TTGACA"], True))
    # a chain member the match list does not decide becomes the level pipeline's frontier, the cascade keeps the roots it has
    cases.append(("left-over contigs share a repeat", cut(base, 2) + [rnd(3000) + rep + rnd(2000)], cut(var, 2) + [rnd(1000) + rep + rnd(4000)], True))
    c3, c4 = cut(base, 9), cut(var, 7)
    rng.shuffle(c4)
    cases.append(("many contigs, shuffled: chance matches between the left-overs", c3, c4, True))
    done = 0
    for k, (what, a, b, want) in enumerate(cases):
        inputs = [_write_fasta(tmp_path / ("a%d.fa" % k), a), _write_fasta(tmp_path / ("b%d.fa" % k), b)]
        info = check(inputs, 20, sa64, want_done=want)
        done += bool(info["done"])
        if want:
            assert info["subindices"] > 100, (what, info)
    assert done == len(cases)
    # the same through the level pipeline (what these inputs took before): identical by the same checks
    monkeypatch.setenv("RV_NO_CASCADE_CHAIN", "1")
    inputs = [_write_fasta(tmp_path / "a.fa", c1), _write_fasta(tmp_path / "b.fa", c2)]
    assert not check(inputs, 20, sa64)["done"]


@pytest.mark.parametrize("sa64", [False, True])
def test_anchors_delivered_into_the_callers_arrays(sa64):
    """rv_set_result_buffers (include/reveal_amd.h): a built-in run on a handle whose previous result arrays have been let go of writes its anchors
    straight into them (page-locked by the library); arrays somebody still holds are left alone.  Same anchors either way, run after run, for results
    that grow, shrink and do not fit"""
    big = [g.decode() for g in synth.genomes(400000, 2, seed=21)]
    small = [g.decode() for g in synth.genomes(90000, 2, seed=22)]
    ref = {id(x): aset(oracle_run(x, 20, sa64)["anchors"]) for x in (big, small)}
    idx_big, idx_small = feed(mod(sa64).index(), big), feed(mod(sa64).index(), small)
    for idx, inp in ((idx_big, big), (idx_small, small)):
        kept = None
        for turn in range(5):
            idx.construct()
            got = idx.align_builtin(20, 2)
            assert aset(got["anchors"]) == ref[id(inp)], (turn, len(inp[0]))
            if turn == 1:
                kept = got["anchors"]            # views of the arrays stay with the caller: the next run must not write into them
                snap = [np.array(a, copy=True) for a in kept]
            elif turn == 2:
                assert all(np.array_equal(a, b) for a, b in zip(kept, snap))
                kept = None
            del got
    # one handle, results of different sizes through the same arrays (the larger one does not fit the smaller one's)
    idx = feed(mod(sa64).index(), small)
    for inp in (small, small, big, big, small, small):
        idx2 = feed(mod(sa64).index(), inp)
        idx2.__dict__["_res_bufs"] = idx.__dict__.get("_res_bufs")
        idx.__dict__.pop("_res_bufs", None)
        idx._dll.rv_set_result_buffers(idx._h, None, 0, None, 0, None, 0)
        idx2.construct()
        got = idx2.align_builtin(20, 2)
        assert aset(got["anchors"]) == ref[id(inp)]
        del got
        idx = idx2
