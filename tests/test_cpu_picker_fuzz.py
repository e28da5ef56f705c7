"""rv_pick_chain (the reference's default picker in C++, include/reveal_amd.h) beside schemes.GraphPicker.graphmumpicker on random match lists:
overlapping matches (trim_overlap's filter and cuts, incl. the places where the reference's own code raises), equal lengths and equal offsets
(the stable sorts, the dictionary that lets a later match replace an earlier one), subsets of the samples (`segment`), seeds above a small
--seedsize, the three gap-cost models, --notrim, minlength 0 (the p-value cut).  No index and no GPU involved: the picker is host code."""
import random

import pytest

from reveal_amd import schemes


class FakeGraph:
    """what the picker reads of an alignment graph whose nodes are whole sequences (FASTA inputs, one sequence per sample)"""

    def __init__(self, seqs):
        self.nodes = list(seqs)
        self.offsets = {n: {k: 0} for k, n in enumerate(seqs)}
        self.id2path = {k: "s%d" % k for k in range(len(seqs))}
        self.path2id = {v: k for k, v in self.id2path.items()}
        self.paths = [self.id2path[k] for k in range(len(seqs))]
        self.id2end = {k: e - b for k, (b, e) in enumerate(seqs)}

    def node_at(self, pos):
        for n in self.nodes:
            if n[0] <= pos < n[1]:
                return n
        raise KeyError(pos)


class FakeIdx:
    def __init__(self, nodes, nsamples, leftnode, rightnode):
        self.nodes, self.nsamples, self.leftnode, self.rightnode, self.depth = nodes, nsamples, leftnode, rightnode, 1


def one_case(rng):
    ns = rng.choice([2, 2, 2, 3, 4])
    L = rng.choice([200, 2000, 40000])
    seqs, at = [], 0
    for _ in range(ns):
        seqs.append((at, at + L)); at += L + 1
    G = FakeGraph(seqs)
    # the sub-index: an interval of every sample (sometimes not of all), its left / right graph nodes standing right outside
    present = [k for k in range(ns) if rng.random() < 0.9] or [0]
    if len(present) < 2:
        present = [0, 1]
    ivs = {}
    for k in present:
        b = seqs[k][0] + rng.randint(0, L // 3); e = seqs[k][1] - rng.randint(0, L // 3)
        ivs[k] = (b, e)
    root = rng.random() < 0.3
    if root:
        ivs = {k: seqs[k] for k in range(ns)}; present = list(range(ns))
        ln = rn = None
    else:
        ln, rn = (-10, -9), (-20, -19)
        G.offsets[ln] = {k: ivs[k][0] - 1 - seqs[k][0] for k in present}      # + (ln[1] - ln[0]) - 1 = the interval's begin - 1
        G.offsets[rn] = {k: ivs[k][1] - seqs[k][0] for k in present}
    idx = FakeIdx(set(ivs.values()), len(present), ln, rn)
    m = rng.choice([1, 2, 3, 8, 30, 120])
    mums = []
    maxl = max(2, min(300, (min(e - b for b, e in ivs.values())) // 2))
    for _ in range(m):
        l = rng.randint(1, maxl)
        if rng.random() < 0.2 and mums:      # equal length / nearly the same place as an earlier one
            l = mums[-1][0]
        members = present if rng.random() < 0.8 else rng.sample(present, rng.randint(2, len(present)))
        d = rng.randint(0, max(0, min(ivs[k][1] - ivs[k][0] for k in members) - l))
        spd = []
        for k in members:
            jitter = rng.choice([0, 0, 0, 1, -1, 5]) if rng.random() < 0.5 else 0
            p = min(max(ivs[k][0] + d + jitter, ivs[k][0]), ivs[k][1] - l)
            spd.append((k, p))
        rng.shuffle(spd)                      # (the scan emits members in suffix-array order)
        mums.append((l, len(spd), tuple(spd)))
    args = schemes.PickerArgs(wscore=rng.choice([1, 1, 3]), wpen=rng.choice([1, 1, 2]), maxmums=rng.choice([1000, 5, 2]), seedsize=rng.choice([10000, 3, 1]),
                              gcmodel=rng.choice(["sumofpairs", "star-avg", "star-med"]), trim=rng.random() < 0.7, pcutoff=rng.choice([1e-8, 0.5, 1.0]))
    minlength = rng.choice([20, 20, 0])
    return G, idx, mums, args, minlength, seqs, ivs


def norm(r):
    if not r:
        return ()
    f = lambda mm: (mm[0], mm[1], tuple(tuple(x) for x in mm[2]))
    return f(r[0]), [(f(mm), sc) for mm, sc in r[1]], [(f(mm), sc) for mm, sc in r[2]]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_native_picker_on_random_lists(seed):
    rng = random.Random(seed)
    agree = raised = picked = 0
    for _ in range(1500):
        G, idx, mums, args, minlength, seqs, ivs = one_case(rng)
        ns = len(seqs)
        try:
            want = schemes.GraphPicker(G, args).graphmumpicker(list(mums), idx, precomputed=False, minlength=minlength)
            err = None
        except (IndexError, KeyError) as e:      # the reference's own code raises on such a list
            want, err = None, e
        ivb = [ivs[k][0] if k in ivs else -1 for k in range(ns)]
        ive = [ivs[k][1] if k in ivs else -1 for k in range(ns)]
        try:
            got = schemes.native_pick(mums, idx.nsamples, [b for b, _ in seqs], ivb, ive, args, minlength)
            gerr = None
        except RuntimeError as e:
            got, gerr = None, e
        if err is not None or gerr is not None:
            assert err is not None and gerr is not None, (seed, mums, err, gerr)
            raised += 1
            continue
        assert norm(got) == norm(want), (seed, mums, args.__dict__, minlength, norm(got), norm(want))
        agree += 1
        picked += 1 if want else 0
    assert agree > 1000 and picked > 300, (agree, raised, picked)
