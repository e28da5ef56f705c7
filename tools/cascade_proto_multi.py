#!/usr/bin/env python3
"""CPU prototype of the anchor cascade for k >= 2 samples (one sequence each), beside the oracle's literal recursion.
Test infrastructure (imports oracle/).

The picker of the benchmark callbacks only takes matches present in EVERY sample of the sub-index (schemes.py:227).  For a
descendant C of the root X that still holds all k samples:
  * R[p] = the longest prefix suffix p shares with another suffix of ITS OWN sample (a repeat inside one genome);
  * a full match of C (k suffixes, one per sample) longer than Rmax(C) = max R over C's positions is an LCP interval of X
    with exactly those k members (an extra member would belong to one of the k samples and raise that member's R), i.e. a
    full match of X cut to C -- shifted behind the matched text in front of C, capped at C's ends -- and every such cut
    match longer than Rmax(C) is a full match of C.
So C's choice is known when its best cut match is longer than Rmax(C); C is empty when it has no cut match of minl and
Rmax(C) < minl, or an interval shorter than minl; a sub-index that lacks a sample, or is not decided, is rebuilt from its text.
usage: python tools/cascade_proto_multi.py [seconds] [seed]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from helpers import assemble, oracle  # noqa: E402

STATS = dict(certain=0, empty=0, rebuilt=0, rebuilt_bases=0, cases=0, total_bases=0, lacking=0)


def oracle_anchors(seqs, minl):
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, len(seqs))
    r = O.align_bench(c, nodes, minl, 2)
    l, n, off, pos = r["anchors"]
    return sorted((int(l[k]), tuple(sorted(int(x) for x in pos[off[k]:off[k + 1]]))) for k in range(len(l)))


def cascade(seqs, minl, depth=0):
    """-> sorted anchors [(l, (p_0 .. p_k-1))], positions in the assembled text of `seqs`"""
    k = len(seqs)
    if k < 2:
        return []
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, k)
    SA, LCP = c["SA"].astype(np.int64), c["LCP"].astype(np.int64)
    n = len(SA)
    so = np.searchsorted(np.asarray(nsep, dtype=np.int64), SA, side="left")      # sample of every rank's suffix
    # R per rank: nearest suffix of the same sample above and below, range minimum of LCP in between
    R = np.zeros(n, dtype=np.int64)
    last = {}
    # upward pass
    for j in range(n):
        s = int(so[j])
        best = 0
        mn = None
        i = j
        while i > 0:
            mn = int(LCP[i]) if mn is None else min(mn, int(LCP[i]))
            if mn < 1:
                break
            i -= 1
            if so[i] == s:
                best = mn
                break
        R[j] = max(R[j], best)
        if best and i >= 0 and so[i] == s:
            R[i] = max(R[i], best)
    Rpos = np.zeros(n, dtype=np.int64)
    Rpos[SA] = R
    if k == 2:
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, minl)
        cand = [((int(a[i]), int(b[i])), int(l[i])) for i in range(len(l))]
    else:
        ml, mn_, moff, mso, mpos = O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, k, minl, 2)
        cand = []
        for i in range(len(ml)):
            if mn_[i] == k:
                cand.append((tuple(sorted(int(x) for x in mpos[moff[i]:moff[i + 1]])), int(ml[i])))
    anchors = []
    stack = [(tuple(nodes), cand)]
    whole = sum(e - b for b, e in nodes)
    while stack:
        iv, cl = stack.pop()
        if sum(e - b for b, e in iv) == 0:
            continue
        lens = [e - b for b, e in iv]
        present = [x > 0 for x in lens]
        if sum(present) < 2:
            STATS["empty"] += 1
            continue
        if not all(present):
            # a sample is missing: the picker now wants matches of the remaining samples -- from the text
            STATS["lacking"] += 1
            if sum(1 for x in lens if x >= minl) < 2:
                STATS["empty"] += 1
                continue
            keep = [s for s in range(k) if present[s]]
            sub = [T[iv[s][0]:iv[s][1]].decode() for s in keep]
            got = cascade(sub, minl, depth + 1) if sum(lens) < whole else oracle_anchors(sub, minl)
            starts = np.cumsum([0] + [len(x) + 1 for x in sub])
            for (ql, ps) in got:      # (an anchor of a deeper sub-index that lacks samples has fewer members: map by position)
                js = [int(np.searchsorted(starts, p, side="right")) - 1 for p in ps]
                anchors.append((ql, tuple(int(p - starts[j] + iv[keep[j]][0]) for j, p in zip(js, ps))))
            continue
        if min(lens) < minl:
            STATS["empty"] += 1
            continue
        rmax = max(int(Rpos[b:e].max()) for b, e in iv)
        best = None
        live = []
        for (ps, ln) in cl:
            sh = max(0, max(iv[s][0] - ps[s] for s in range(k)))
            ql = ln - sh
            ql = min(ql, min(iv[s][1] - (ps[s] + sh) for s in range(k)))
            if ql < minl or any(ps[s] + sh < iv[s][0] for s in range(k)):
                continue
            live.append((ps, ln))
            q0 = ps[0] + sh
            if best is None or ql > best[1] or (ql == best[1] and q0 < best[0][0]):
                best = (tuple(p + sh for p in ps), ql)
        if best is not None and best[1] > rmax:
            STATS["certain"] += 1
            qs, ql = best
            anchors.append((ql, qs))
            stack.append((tuple((iv[s][0], qs[s]) for s in range(k)), live))
            stack.append((tuple((qs[s] + ql, iv[s][1]) for s in range(k)), live))
            continue
        if best is None and rmax < minl:
            STATS["empty"] += 1
            continue
        STATS["rebuilt"] += 1
        STATS["rebuilt_bases"] += sum(lens)
        sub = [T[b:e].decode() for b, e in iv]
        if depth > 40:
            raise RuntimeError("no progress")
        got = cascade(sub, minl, depth + 1) if sum(lens) < whole else oracle_anchors(sub, minl)
        starts = np.cumsum([0] + [len(x) + 1 for x in sub])
        for (ql, ps) in got:
            js = [int(np.searchsorted(starts, p, side="right")) - 1 for p in ps]
            anchors.append((ql, tuple(int(p - starts[j] + iv[j][0]) for j, p in zip(js, ps))))
    return sorted(anchors)


def main():
    from fuzz import make_case
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t_end = time.time() + budget
    bad = 0
    while time.time() < t_end:
        seqs, minl = make_case(rng)
        if len(seqs[0]) > 30000:          # (the R pass above is a Python loop)
            seqs = [s[:30000] for s in seqs]
        if any(len(s) == 0 for s in seqs):
            continue
        want = oracle_anchors(seqs, minl)
        got = cascade(seqs, minl)
        STATS["cases"] += 1
        STATS["total_bases"] += sum(len(s) for s in seqs)
        if got != want:
            bad += 1
            print("MISMATCH case", STATS["cases"], "k", len(seqs), "minl", minl, "lens", [len(s) for s in seqs], "anchors", len(want), len(got))
            sw, sg = set(want), set(got)
            print("  only oracle:", sorted(sw - sg)[:4], " only cascade:", sorted(sg - sw)[:4])
            if bad >= 3:
                break
    print(STATS, "mismatches", bad)


if __name__ == "__main__":
    main()
