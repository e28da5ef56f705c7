"""The anchor cascade's decision rules on the CPU, beside the oracle's literal recursion (reveal.c:731-1338 with the benchmark
callbacks): tools/cascade_proto.py (two samples, bound W) and tools/cascade_proto_multi.py (two to four samples, bound R = repeats
inside one sample) restate what reveal_amd/csrc/rv_cascade.hip / rv_cascade_multi.hip decide on the device -- which sub-indices are
split on which match, which are empty, which are rebuilt from their text -- and must give the oracle's anchors on random inputs
with substitutions, indels, tandem repeats, N runs and identical copies.  (The long soaks are in the tools' own main().)"""
import os
import random
import sys

import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def cases(seed, count, max_len, samples):
    from fuzz import make_case
    rng = random.Random(seed)
    out = []
    while len(out) < count:
        seqs, minl = make_case(rng)
        seqs = [s[:max_len] for s in seqs[:samples[1]]]
        if len(seqs) < samples[0] or any(len(s) == 0 for s in seqs):
            continue
        out.append((seqs, minl))
    return out


def test_two_sample_rules_give_the_oracles_anchors():
    import cascade_proto as P
    for seqs, minl in cases(101, 40, 20000, (2, 2)):
        want = P.oracle_anchors(seqs, minl)[0]
        assert P.cascade(seqs, minl) == want
    assert P.STATS["certain"] > 100 and P.STATS["rebuilt"] > 0


def test_multi_sample_rules_give_the_oracles_anchors():
    import cascade_proto_multi as P
    for seqs, minl in cases(202, 40, 6000, (2, 4)):
        assert P.cascade(seqs, minl) == P.oracle_anchors(seqs, minl)
    assert P.STATS["certain"] > 100 and P.STATS["rebuilt"] + P.STATS["lacking"] > 0
