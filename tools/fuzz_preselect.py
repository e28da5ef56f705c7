#!/usr/bin/env python3
"""Randomised soak of the anchor pre-selection (index.preselect / rv_set_preselect) on the random inputs
of tools/fuzz.py: align() with Python callbacks that start like schemes.graphmumpicker (filter, two stable sorts, cap) and
then pick by the whole capped list -- with pre-selection on -- against the same callbacks driving the CPU oracle alone
(tests/test_gpu_preselect.py's oracle_recursion): callback trace, anchors, final text.  Test infrastructure.
usage: python tools/fuzz_preselect.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz import make_case  # noqa: E402
from test_gpu_preselect import oracle_recursion, run  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    t0, n, saved = time.time(), 0, [0, 0]
    while time.time() - t0 < budget:
        seqs, minl = make_case(rng)
        if max(len(s) for s in seqs) > 120000:
            continue
        seqs = [s.decode() if isinstance(s, (bytes, bytearray)) else s for s in seqs]
        sa64 = rng.random() < 0.25
        maxmums = rng.choice([1, 2, 5, 40, 1000])
        os.environ["RV_PRESEL_DEV_MIN"] = rng.choice(["0", "65536"])      # two samples: the cap applied on the device at every level / on the host
        try:
            T1, an1, tr1, _ = run(seqs, maxmums, True, minl, sa64)
            To, ano, tro = oracle_recursion(seqs, maxmums, minl, sa64)
            tr1, tro = ([r for r in t if r[2] > 0] for t in (tr1, tro))
            assert T1 == To and an1 == ano
            assert [(d, b, rel) for d, b, _, rel in tr1] == [(d, b, rel) for d, b, _, rel in tro]
            saved[0] += sum(r[2] for r in tr1)
            saved[1] += sum(r[2] for r in tro)
        except Exception:
            print("FAILED case %d (seed %d): %d samples, lengths %s, minl %d, sa64 %s, maxmums %d, RV_PRESEL_DEV_MIN %s" % (n, seed, len(seqs), [len(s) for s in seqs], minl, sa64, maxmums, os.environ["RV_PRESEL_DEV_MIN"]))
            raise
        n += 1
    print("fuzz_preselect: %d cases identical to the oracle-driven recursion (seed %d); %d of %d matches crossed into Python" % (n, seed, saved[0], saved[1]))


if __name__ == "__main__":
    main()
