#!/usr/bin/env python3
"""Randomised parity soak of splitindex / extract (reveal.c:1386-1748) against the CPU oracle, on the inputs of
tools/fuzz.py (SNPs, indels, tandem repeats, N runs, 2-4 samples).  Test infrastructure.
usage: python tools/fuzz_steps.py [seconds] [seed]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from helpers import assemble, feed, oracle  # noqa: E402
from fuzz import make_case                    # noqa: E402
from test_gpu_single_steps import drive       # noqa: E402
from reveal_amd import reveallib, reveallib64  # noqa: E402


def random_intervals(rng, nodes, n_iv, maxlen):
    """disjoint, non-touching intervals inside the sequences, ascending"""
    out = []
    for b, e in nodes:
        at = b + rng.randint(0, 5)
        while at + 2 < e and len(out) < n_iv:
            ln = rng.randint(1, maxlen)
            if at + ln >= e:
                break
            out.append((at, at + ln))
            at += ln + rng.randint(1, max(2, (e - b) // max(n_iv, 1)))
    return out


def extract_case(rng, seqs, sa64):
    T, nsep, nodes = assemble(seqs)
    O = oracle(sa64)
    c = O.construct(T, nsep, len(seqs))
    idx = feed((reveallib64 if sa64 else reveallib).index(), seqs)
    idx.construct()
    sa, lcp = c["SA"], c["LCP"]
    for rnd in range(rng.randint(1, 3)):
        # what is still in the index: positions of its SA
        alive = np.zeros(len(T) + 1, dtype=bool)
        alive[sa] = True
        ivs = [iv for iv in random_intervals(rng, nodes, rng.randint(1, 40), rng.choice([3, 30, 300]))
               if alive[iv[0]:iv[1]].all() and not (iv[0] <= sa[0] < iv[1])]
        # keep them apart from what was extracted before (a hole inside or next to an interval is "not part of this index")
        if not ivs:
            break
        # (any order is allowed only for intervals farther apart than the longest repeat: identical samples of 250 kbp have one of 250 kbp)
        far = max(100000, idx.maxlcp)
        rng.shuffle(ivs) if rng.random() < 0.3 and all(abs(a[0] - b[0]) > far for a in ivs for b in ivs if a != b) else None
        sa, lcp, _ = O.extract(c["tbuf"], sa, lcp, c["SAi"], c["nsep"], ivs, nT=len(T))
        idx.extract(list(ivs))
        assert idx.n == len(sa), ("n", idx.n, len(sa))
        assert np.array_equal(idx.array("SA"), sa) and np.array_equal(idx.array("LCP"), lcp), "arrays after extract round %d" % rnd
        assert idx.T.encode("latin-1") == bytes(c["tbuf"][:len(T)])
    if len(seqs) == 2:
        l, a, b = O.getmums(c["tbuf"], sa, lcp, c["nsep"], 15, nT=len(T))
        assert idx.getmums(15) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    t0 = time.time()
    n = 0
    while time.time() - t0 < budget:
        seqs, minl = make_case(rng)
        if max(len(s) for s in seqs) > 300000:
            continue
        sa64 = rng.random() < 0.25
        try:
            drive(seqs, minl=minl, sa64=sa64, max_steps=rng.choice([5, 30, 200]))
            extract_case(rng, seqs, sa64)
        except Exception:
            print("FAILED case %d (seed %d): %d samples, lengths %s, minl %d, sa64 %s" % (n, seed, len(seqs), [len(s) for s in seqs], minl, sa64))
            raise
        n += 1
    print("fuzz_steps: %d cases (splitindex-driven recursion + repeated extract) identical to the oracle (seed %d)" % (n, seed))


if __name__ == "__main__":
    main()
