#!/usr/bin/env python3
"""How the anchor cascade fares on inputs that look more like a pair of bacterial strains than the benchmark's uniform text:
interspersed repeats (IS elements), rRNA-like operons, indels, tandem arrays, an inversion, strain-specific insertions.
Prints, per scenario, what the cascade did and the time of construct + recursion; parity against the level pipeline.
usage (GPU box): python tools/realistic_probe.py [L]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reveal_amd import reveallib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
rng = random.Random(5)
nrng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(n):
    return ACGT[nrng.integers(0, 4, n)].tobytes().decode()


def mutate(s, snp=0.01, indel=0.0):
    a = np.frombuffer(s.encode(), dtype=np.uint8).copy()
    idx = nrng.random(len(a)) < snp
    a[idx] = ACGT[nrng.integers(0, 4, int(idx.sum()))]
    s = a.tobytes().decode()
    if indel > 0:
        out, i = [], 0
        cuts = sorted(rng.sample(range(len(s)), int(len(s) * indel)))
        for c in cuts:
            out.append(s[i:c])
            if rng.random() < 0.5:
                i = min(len(s), c + rng.randint(1, 40))
            else:
                out.append(rand_seq(rng.randint(1, 40))); i = c
        out.append(s[i:])
        s = "".join(out)
    return s


def insert_copies(s, unit, copies):
    pos = sorted(rng.sample(range(len(s)), copies))
    out, i = [], 0
    for p in pos:
        out.append(s[i:p]); out.append(unit); i = p
    out.append(s[i:])
    return "".join(out)


def scenarios():
    base = rand_seq(L)
    yield "uniform + 1 % SNP (the benchmark's kind)", base, mutate(base)
    yield "+ 0.05 % indels", base, mutate(base, indel=0.0005)
    b2 = insert_copies(base, rand_seq(1500), 20)
    yield "20 copies of a 1.5 kb element in the ancestor", b2, mutate(b2)
    b3 = insert_copies(base, rand_seq(5000), 7)
    yield "7 copies of a 5 kb operon", b3, mutate(b3, indel=0.0002)
    unit = rand_seq(37)
    b4 = base[:L // 3] + unit * 60 + base[L // 3:]
    yield "a 2.2 kb tandem array (diverged copies)", b4[:L // 3] + mutate(unit * 60, 0.02) + b4[L // 3 + 60 * 37:], mutate(b4)
    b5 = mutate(base)
    inv = b5[L // 2:L // 2 + 30000][::-1].translate(str.maketrans("ACGT", "TGCA"))
    yield "a 30 kb inversion + strain-specific 40 kb insertion", base, b5[:L // 2] + inv + b5[L // 2 + 30000:3 * L // 4] + rand_seq(40000) + b5[3 * L // 4:]


def run(a, b, nocascade):
    if nocascade:
        os.environ["RV_NO_CASCADE"] = "1"
    else:
        os.environ.pop("RV_NO_CASCADE", None)
    idx = reveallib.index()
    idx.addsample("a"); idx.addsequence(a)
    idx.addsample("b"); idx.addsequence(b)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        idx.construct()
        res = idx.align_builtin(20, 2)
        best = min(best, time.perf_counter() - t0)
    l, off, pos = res["anchors"]
    return best * 1e3, idx.cascade_info(), sorted(zip(l.tolist(), pos[0::2].tolist(), pos[1::2].tolist())), idx.T


for name, a, b in scenarios():
    t1, info, an1, T1 = run(a, b, False)
    t0, _, an0, T0 = run(a, b, True)
    print("%-52s cascade %7.2f ms | level pipeline %7.2f ms | same anchors %s text %s | done %s levels %d undecided %d rebuilt %d anchors %d" % (
        name, t1, t0, an1 == an0, T1 == T0, info["done"], info["levels"], info["undecided"], info["rebuilt_ranks"], len(an1)), flush=True)
