#!/usr/bin/env python3
"""Randomised parity soak: the GPU recursion (traced and untraced, default and forced code paths) against the CPU oracle
on random inputs -- SNPs, indels, tandem repeats, N runs, several contigs, 2-4 samples.  Test infrastructure.
usage: python tools/fuzz.py [seconds] [seed]      (FUZZ_BIG=1 adds inputs of 1 and 2.5 Mbp per sample)"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assemble, feed, oracle  # noqa: E402
from reveal_amd import reveallib, reveallib64, shard  # noqa: E402

VERBOSE = bool(os.environ.get("FUZZ_VERBOSE"))      # print every configuration before it runs (to find the case behind a GPU fault)
ONLY = int(os.environ.get("FUZZ_ONLY", "-1"))
FIELDS = ("key", "n", "depth", "nsamples", "nnodes", "nmums", "picked", "l", "mn", "sp_min", "h_sa", "h_lcp", "h_mums")
ENVS = [
    {},                                                   # defaults: the anchor cascade decides untraced two-sample runs (rv_cascade.hip)
    {"RV_NO_CASCADE": "1"},                               # ... the level pipeline for them too
    {"RV_BUBBLE_PAR_MIN": "0", "RV_NO_LEAF": "1"},        # (RV_NO_LEAF also keeps the cascade off)
    {"RV_BUBBLE_PAR_MIN": "256", "RV_NO_CASCADE": "1"},
    {"RV_BUBBLE_LDS_ALWAYS": "1", "RV_NO_LEAF": "1"},
    {"RV_NO_EARLY_SPLIT": "1", "RV_NO_CASCADE": "1"},
    {"RV_CARRY_CH": "2", "RV_SA_NO_TEXT": "1", "RV_LCP_BY_RANK": "1", "RV_NO_CASCADE": "1"},
    {"RV_BUBBLE_PAR_MIN": "64", "RV_PB_TWO_PASS": "1", "RV_NO_LEAF": "1"},
    {"RV_LEAF_ACAP": "2", "RV_NO_CASCADE": "1"},
    {"RV_LEAF_ACAP": "2"},                                # the cascade's leaf launch with a two-anchor staging area
    {"RV_CASCADE_SECOND": "2"},                           # two samples through the interval cascade (rv_cascade_multi.hip) as well
    {"RV_CASM_RANK_COUNT": "1"},                          # more than two samples: undecided sub-indices ranked by counting instead of sorting (k_casm_rank)
    {"RV_CASM_BIG_MIN": "0"},                             # ... every one of them rebuilt through global memory (k_casmb_*: the path of those above 8192 ranks)
    {"RV_CASCADE_DANGER": "2"},                           # every undecided sub-index decided from its witnesses where they can be (k_cas_dwalk)
    {"RV_CASCADE_DANGER": "2", "RV_CASCADE_DANGER_MIN": "300", "RV_CASCADE_SECOND_OFF": "1"},
    {"RV_CASCADE_DANGER": "2", "RV_NO_DWALK_BLOCKS": "1"},   # ... every witness walks in global memory (no blocks in LDS)
    {"RV_NO_TWIN_COLLAPSE": "1"},                         # every suffix of the second sample through the radix sort
    {"RV_DIAG_TABLE": "1"},                               # two samples: hint and twins along piecewise diagonals from seeds, whatever the input
    {"RV_NO_SHORT_ALPHABET": "1"},                        # a digit value of its own for "past the end"
    {"RV_NO_TINY_SA": "1"},                               # texts of up to 2048 characters through the general build, too
    {"RV_FAR_TABLE": "1"},                                # the far round reads the next mark of a pair from the text-order table (k_far_nd) whatever the list's size
    {"RV_NO_FAR_TWINS": "1"},                             # ties beyond the text round: doubling rounds for partners, too (round 5)
    {"RV_NO_SLOW_CLASS": "1"},                            # ... every tied entry through every doubling round
    {"RV_NO_LCP_LIST": "1", "RV_NO_TEXT_JUMP": "1"},      # ... and the whole index through rv_build_lcp after them
    {"RV_SCAN_V1": "1"},                                  # round 5's staged multi-sample scans (k_casm_scan, k_multi_pick1) instead of k_full_scan
    {"RV_PICK_THREADS": "1"},                             # (no effect on the built-in picker: the switch must at least be accepted)
]


def mutate(rng, base, snp, indel):
    out = []
    i = 0
    L = len(base)
    while i < L:
        r = rng.random()
        if r < snp:
            out.append(rng.choice("ACGT")); i += 1
        elif r < snp + indel:
            if rng.random() < 0.5:
                i += rng.randint(1, 30)
            else:
                out.append("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 30))))
        else:
            out.append(base[i]); i += 1
    return "".join(out)


def make_case(rng):
    L = rng.choice([300, 2000, 20000, 80000, 250000] + ([1000000, 2500000] if os.environ.get("FUZZ_BIG") else []))
    ns = rng.choice([2, 2, 2, 3, 4])
    base = "".join(rng.choice("ACGT") for _ in range(L))
    if rng.random() < 0.4:      # tandem repeats / low complexity
        p = rng.randint(0, L - 1); unit = base[p:p + rng.randint(1, 40)] or "A"
        base = base[:p] + unit * rng.randint(3, 200) + base[p:]
    if rng.random() < 0.3:      # an N run
        p = rng.randint(0, len(base) - 1)
        base = base[:p] + "N" * rng.randint(1, 400) + base[p:]
    snp = rng.choice([0.0, 0.001, 0.01, 0.05])
    indel = rng.choice([0.0, 0.0, 0.0005, 0.003])
    seqs = [base] + [mutate(rng, base, snp, indel) for _ in range(ns - 1)]
    if rng.random() < 0.2:      # identical copy
        seqs[-1] = seqs[0]
    return seqs, rng.choice([10, 20, 20, 30])


def maybe_contigs(rng, seqs):
    """draft assemblies: with two samples, sometimes every sample is cut into contigs (one '$' each, utils.py:325-350), the second one's in
    another order, now and then with a contig that has no partner"""
    if len(seqs) != 2 or rng.random() >= 0.35:
        return seqs

    def cut(s):
        k = rng.randint(1, 6)
        at = sorted(rng.sample(range(1, len(s)), min(k - 1, len(s) - 1))) if len(s) > 2 else []
        return [s[i:j] for i, j in zip([0] + at, at + [len(s)])]
    a, b = cut(seqs[0]), cut(seqs[1])
    rng.shuffle(b)
    if rng.random() < 0.3:
        a.append("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 3000))))
    return [a, b]


def digest(tr):
    o = np.lexsort((tr["key"], tr["depth"]))
    return {f: tr[f][o].astype(np.uint64) for f in FIELDS}


def anchors_set(a):
    if len(a) == 4:
        l, n, off, pos = a
    else:
        l, off, pos = a
    return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))


def divided(M, seqs, minl, stop, nparts, trace):
    """one alignment over `nparts` handles (frontier hand-off, reveal_amd/shard.py), in this process"""
    os.environ["RV_NO_CASCADE"] = "1"      # (the cascade would finish a two-sample run before there is a frontier to divide)
    owner = feed(M.index(), seqs)
    owner.construct()
    lib = owner._lib
    left = owner.align_builtin_until(stop, minl, 2, trace=trace)
    res = []
    if left > 0:
        fr = owner.frontier()
        packed = []
        for subs in shard.partition(fr["meta"][:, 1], nparts):
            part = shard.subset(fr, subs)
            m = int(part["meta"][:, 1].sum())
            bufs = (np.zeros(max(m, 1), lib.sa_t), np.zeros(max(m, 1), lib.lcp_t), np.zeros(max(m, 1), np.uint8))
            owner.frontier_pack(subs, *bufs)
            packed.append((part, bufs))
        owner.frontier_import(packed[0][0], *packed[0][1], minl=minl, minn=2)
        for part, bufs in packed[1:]:
            if len(part["meta"]) == 0:
                continue
            w = feed(M.index(), seqs)
            w.frontier_import(part, *bufs, minl=minl, minn=2, maxlcp=owner.maxlcp, trace=trace)
            res.append(w.align_builtin_resume())
    res.insert(0, owner.align_builtin_resume())
    return shard.merge(res)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    t_end = time.time() + budget
    ncase = 0
    while time.time() < t_end:
        seqs, minl = make_case(rng)
        seqs = maybe_contigs(rng, seqs)
        sa64 = rng.random() < 0.2
        if ONLY >= 0 and ncase != ONLY:      # replay the generator up to one case (FUZZ_ONLY=<case>)
            for _ in range(2):
                rng.choice([2, 3, 8, 32]); rng.choice([2, 3, 5])
            ncase += 1
            if ncase > ONLY:
                break
            continue
        T, nsep, nodes = assemble(seqs)
        O = oracle(sa64)
        c = O.construct(T, nsep, len(seqs))
        sa_ref, lcp_ref = c["SA"].copy(), c["LCP"].copy()
        ref = O.align_bench(c, nodes, minl, 2, trace_cap=4 * len(T) // max(minl, 1) + 1000)
        rd, ra = digest(ref["trace"]), anchors_set(ref["anchors"])
        for env in ENVS:
            for k in list(os.environ):
                if k.startswith("RV_"):
                    del os.environ[k]
            os.environ.update(env)
            for trace in (True, False):
                idx = feed((reveallib64 if sa64 else reveallib).index(), seqs)
                idx.construct()
                tag = "seed %d case %d env %s trace %s sa64 %s (L %d, %d samples, minl %d)" % (seed, ncase, env, trace, sa64, sum(len(c) for c in seqs[0]) if isinstance(seqs[0], list) else len(seqs[0]), len(seqs), minl)
                if VERBOSE:
                    print(tag, file=sys.stderr, flush=True)
                assert np.array_equal(idx.array("SA"), sa_ref), "SA " + tag
                assert np.array_equal(idx.array("LCP"), lcp_ref), "LCP " + tag
                got = idx.align_builtin(minl, 2, trace=trace)
                assert anchors_set(got["anchors"]) == ra, "anchors " + tag
                assert idx.T.encode("latin-1") == ref["T"], "text " + tag
                if trace:
                    gd = digest(got["trace"])
                    for f in FIELDS:
                        assert len(gd[f]) == len(rd[f]) and (gd[f] == rd[f]).all(), "trace field %s %s" % (f, tag)
        # the same alignment divided over several handles at a random level (default code paths)
        for k in list(os.environ):
            if k.startswith("RV_"):
                del os.environ[k]
        for trace in (True, False):
            stop, nparts = rng.choice([2, 3, 8, 32]), rng.choice([2, 3, 5])
            tag = "seed %d case %d divided stop %d parts %d trace %s sa64 %s" % (seed, ncase, stop, nparts, trace, sa64)
            if VERBOSE:
                print(tag, file=sys.stderr, flush=True)
            got = divided(reveallib64 if sa64 else reveallib, seqs, minl, stop, nparts, trace)
            assert anchors_set(got["anchors"]) == ra, "anchors " + tag
            assert shard.lower_text(T, got["anchors"]).tobytes() == ref["T"], "text " + tag
            if trace:
                gd = digest(got["trace"])
                for f in FIELDS:
                    assert len(gd[f]) == len(rd[f]) and (gd[f] == rd[f]).all(), "trace field %s %s" % (f, tag)
        ncase += 1
    print("fuzz: %d cases x (%d configurations + divided run) x 2 (traced / untraced) identical to the oracle (seed %d)" % (ncase, len(ENVS), seed))


if __name__ == "__main__":
    main()
