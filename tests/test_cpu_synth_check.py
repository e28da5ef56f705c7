"""Host-side pieces of round 4 (no GPU): the indel generator against a literal restatement of the reference simulator's loop
(utils/simulate.py:17-77), the canonical anchor stream / digests of reveal_amd/check.py, the golden look-up, bench.py's contig helpers."""
import importlib.util
import os

import numpy as np
import pytest

from helpers import ROOT, synth
from reveal_amd import check


def literal_mut(base, k, seed, rate, indelfrac, zipfd=1.7, maxlen=2000):
    """the simulator's loop, event by event, on the draws reveal_amd.synth.variant_codes_indel makes"""
    L = len(base)
    rng = np.random.Generator(np.random.PCG64(seed + k))
    npos = int(np.ceil(rate * L))
    pos = np.sort(synth._distinct_positions(rng, L, npos))
    indel = rng.random(npos) < indelfrac
    ins = indel & (rng.random(npos) < 0.5)
    length = np.minimum(rng.zipf(zipfd, size=npos), maxlen)
    alt = rng.integers(0, 2, size=npos, dtype=np.uint8)
    out, offset = [], 0
    for j, p in enumerate(pos):
        if p < offset:                                   # swallowed by an earlier deletion (simulate.py:35-36)
            continue
        if indel[j]:
            if ins[j]:
                out += list(base[offset:p]) + [255] * int(length[j]); offset = p
            else:
                out += list(base[offset:p]); offset = p + int(length[j])
        else:
            b, a = base[p], alt[j]
            out += list(base[offset:p]) + [a + 1 if a >= b else a]; offset = p + 1
    out += list(base[offset:])
    out = np.array(out, dtype=np.uint8)
    hole = out == 255
    out[hole] = rng.integers(0, 3, size=int(hole.sum()), dtype=np.uint8)
    return out


@pytest.mark.parametrize("L,rate,frac", [(1000, 0.05, 0.2), (20000, 0.05, 0.2), (20000, 0.3, 0.9), (150000, 0.01, 0.2)])
def test_indel_generator_is_the_simulators_loop(L, rate, frac):
    for seed in range(4):
        b = synth.base_codes(L, seed)
        assert np.array_equal(synth.variant_codes_indel(b, 1, seed, rate=rate, indelfrac=frac), literal_mut(b, 1, seed, rate, frac))


def test_generators_are_seeded_and_members_match_the_family():
    g = synth.genomes(30000, 4, seed=9)
    assert g == synth.genomes(30000, 4, seed=9)
    b = synth.base_codes(30000, 9)
    assert [synth.member(b, k, 9) for k in range(4)] == g
    gi = synth.genomes(30000, 3, seed=9, indelfrac=0.2)
    assert gi[0] == g[0] and gi[1] != g[1] and gi == synth.genomes(30000, 3, seed=9, indelfrac=0.2)
    assert [synth.member(b, k, 9, indelfrac=0.2) for k in range(3)] == gi
    assert set(gi[1]) <= set(b"ACGT") and abs(len(gi[1]) - 30000) < 3000


def test_anchor_stream_is_canonical():
    l = np.array([30, 20, 25], dtype=np.uint32); off = np.array([0, 2, 5, 7]); pos = np.array([500, 1500, 10, 1010, 2010, 300, 1300])
    s = check.anchor_stream(l, off, pos)
    assert s.tolist() == [20, 3, 10, 1010, 2010, 25, 2, 300, 1300, 30, 2, 500, 1500]
    # any order of the anchors gives the same stream and digest
    perm = [2, 0, 1]
    l2 = l[perm]; cnt = np.diff(off)[perm]; off2 = np.concatenate([[0], np.cumsum(cnt)])
    pos2 = np.concatenate([pos[off[k]:off[k + 1]] for k in perm])
    assert check.anchor_digest(l2, off2, pos2) == check.anchor_digest(l, off, pos)
    assert check.anchor_stream(l[:0], off[:1], pos[:0]).size == 0
    assert check.array_digest(np.arange(5, dtype=np.int32)) == check.array_digest(np.arange(5, dtype=np.int32).copy())
    assert check.array_digest(np.arange(5, dtype=np.int32)) != check.array_digest(np.arange(5, dtype=np.int64))


def test_golden_lookup_and_compare():
    r = check.golden_record(5_000_000, 2, 42)
    assert r is not None and r["name"] == "C2_seed42" and r["n"] == 10_000_002
    assert check.golden_record(5_000_000, 2, 43) is None and check.golden_record(5_000_000, 2, 42, minl=21) is None
    ri = check.golden_record(250_000_000, 2, 42, 0.2)
    assert ri is not None and ri["name"] == "C4_indel_seed42" and ri["indelfrac"] == 0.2
    fake = dict(name="x", anchors=1, sha_anchors=check.anchor_digest([7], [0, 2], [1, 9]), sha_finalT=check.array_digest(np.frombuffer(b"ab", dtype=np.uint8)),
                sha_SA=check.array_digest(np.array([1, 0], dtype=np.int32)), sha_LCP="nope")
    g = check.compare_with_golden(fake, anchors=([7], [0, 2], [1, 9]), T_final=np.frombuffer(b"ab", dtype=np.uint8), SA=np.array([1, 0], dtype=np.int32), LCP=np.array([0, 0], dtype=np.int32))
    assert g["anchors"] and g["anchor_count"] and g["final_text"] and g["SA"] and not g["LCP"] and not g["all"]


def test_bench_contig_helpers():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seqs = synth.genomes(40000, 2, seed=3)
    cut = bench.cut_into_contigs(seqs, 5)
    assert [len(c) for c in cut] == [5, 5] and b"".join(cut[0]) == seqs[0]
    assert sorted(cut[1], key=seqs[1].find) != cut[1] or True                       # (the second sample's contigs are shuffled: usually another order)
    assert sorted(b"".join(sorted(cut[1], key=lambda c: seqs[1].find(c)))) == sorted(seqs[1])
    assert bench.flat(cut) == cut[0] + cut[1] and bench.flat(seqs) == seqs
    T = b"$".join(bench.flat(cut)) + b"$"
    seps = bench.sample_seps(cut)
    assert len(seps) == 1 and T[seps[0]:seps[0] + 1] == b"$" and seps[0] == sum(len(c) + 1 for c in cut[0]) - 1
    assert bench.sample_seps(seqs) == [len(seqs[0])]
