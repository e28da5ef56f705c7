#!/usr/bin/env python3
"""CPU prototype of the anchor cascade (reveal_amd/csrc/rv_cascade.hip): the built-in recursion of a two-sample
alignment decided from the TOP-LEVEL match list wherever that is provably what the reference's recursion
(reveal.c:731-1338 with the bench callbacks) would do, checked against the oracle's literal recursion.
Test infrastructure (imports oracle/).

Facts used (X = a sub-index with one interval per sample, arrays canonical):
  * a gap i (ranks i-1, i) is a UNIQUE CROSS PAIR (ucp) iff LCP[i] > LCP[i-1], LCP[i] > LCP[i+1] and the two
    suffixes lie on different sides of nsep[0]; MUMs of X = left-maximal ucp gaps (reveal.c:61-85).
  * W[p], p = SA[j]: the longest match suffix p has with any suffix of X other than its ucp partner
    = max(LCP[j-1], LCP[j+1]) if gap j is ucp, max(LCP[j], LCP[j+2]) if gap j+1 is ucp, else max(LCP[j], LCP[j+1]).
  * a descendant C of X: every MUM of C longer than Wmax(C) = max W over C's positions is a MUM of X cut to C's
    intervals (shifted to start behind the matched text in front of C, capped at C's ends), and every such cut
    candidate longer than Wmax(C) is a MUM of C.  So C's pick is known whenever its best candidate is longer than
    Wmax(C), and C has no match at all when it has no candidate of minl and Wmax(C) < minl.
  * otherwise C is rebuilt from its text (its arrays only depend on the text of its intervals).
usage: python tools/cascade_proto.py [seconds] [seed]"""
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

DANGER_CAP = int(os.environ.get("DANGER_CAP", "4096"))
STATS = dict(certain=0, empty=0, rebuilt=0, rebuilt_bases=0, cases=0, total_bases=0)


def oracle_anchors(seqs, minl):
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, len(seqs))
    r = O.align_bench(c, nodes, minl, 2)
    l, n, off, pos = r["anchors"]
    return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l))), c, T, nsep, nodes


def cascade(seqs, minl, depth=0):
    """-> sorted anchors [(l, (a, b))] of the two-sequence alignment, positions in the assembled text"""
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, 2)
    SA, LCP = c["SA"].astype(np.int64), c["LCP"].astype(np.int64)
    n = len(SA)
    sep = nsep[0]
    side = SA > sep
    L0 = np.concatenate([LCP, [0, 0]])              # LCP[n] = LCP[n+1] = 0
    Lm1 = np.concatenate([[0], LCP])[:n]            # LCP[j-1]
    gap_ucp = np.zeros(n + 2, dtype=bool)
    j = np.arange(1, n)
    gap_ucp[1:n] = (LCP[j] > LCP[j - 1]) & (LCP[j] > L0[j + 1]) & (side[j] != side[j - 1]) & (LCP[j] > 0)
    jj = np.arange(n)
    W = np.where(gap_ucp[jj], np.maximum(Lm1, L0[jj + 1]), np.where(gap_ucp[jj + 1], np.maximum(LCP, L0[jj + 2]), np.maximum(LCP, L0[jj + 1])))
    Wpos = np.zeros(n, dtype=np.int64)
    Wpos[SA] = W
    l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, minl)
    cand = [(int(a[k]), int(b[k]), int(l[k])) for k in range(len(l))]
    (a0, a1), (b0, b1) = nodes
    anchors = []
    stack = [(a0, a1, b0, b1, cand)]
    while stack:
        a0, a1, b0, b1, cl = stack.pop()
        if a1 - a0 < 1 or b1 - b0 < 1:
            continue
        wmax = max(int(Wpos[a0:a1].max()), int(Wpos[b0:b1].max()))
        best = None
        live = []
        for (pa, pb, ln) in cl:
            k = max(0, a0 - pa, b0 - pb)
            qa, qb, ql = pa + k, pb + k, ln - k
            ql = min(ql, a1 - qa, b1 - qb)
            if ql < minl or qa < a0 or qb < b0:
                continue
            live.append((pa, pb, ln))
            if best is None or ql > best[2] or (ql == best[2] and qa < best[0]):
                best = (qa, qb, ql)
        if best is not None and best[2] > wmax:
            STATS["certain"] += 1
            qa, qb, ql = best
            anchors.append((ql, (qa, qb)))
            stack.append((a0, qa, b0, qb, live))
            stack.append((qa + ql, a1, qb + ql, b1, live))
            continue
        if best is None and wmax < minl:
            STATS["empty"] += 1
            continue
        # not decided by the list.  With ell = the best cut match's length (minl when there is none) and D = {p in C: W[p] >= ell}:
        #   * a match of C of ell or more that is no cut match has both its suffixes in D (a third suffix of X shares its string);
        #   * a cut match of length ell whose suffixes are not in D is a match of C (nothing else in X shares ell characters with them);
        #     one whose suffixes are in D may have a second occurrence inside C -- but then that occurrence is in D as well;
        #   * so among D's suffixes, cut at C's ends, the gaps of ell or more are the gaps of C's own index, and C's choice is the best
        #     of what a scan of D finds and the cut match if it is outside D.  Nothing found and a cut match inside D: not decided.
        ell = best[2] if best is not None else minl
        D = [int(x) for x in np.nonzero(Wpos[a0:a1 - ell + 1] >= ell)[0] + a0] + [int(x) for x in np.nonzero(Wpos[b0:b1 - ell + 1] >= ell)[0] + b0]
        for k_ in ("danger_nodes", "danger_max", "danger_hidden", "danger_sum", "danger_fake"):
            STATS.setdefault(k_, 0)
        if len(D) <= DANGER_CAP:
            STATS["danger_nodes"] += 1; STATS["danger_sum"] += len(D); STATS["danger_max"] = max(STATS["danger_max"], len(D))
            cap = lambda p: T[p:(a1 if p < a1 else b1)]
            D.sort(key=lambda p: (cap(p), p))
            g = [0] * (len(D) + 1)
            for i in range(1, len(D)):
                x, y = cap(D[i - 1]), cap(D[i])
                m = min(len(x), len(y)); k = 0
                while k < m and x[k] == y[k] and x[k] not in b"N$":
                    k += 1
                g[i] = k
            found = None
            if best is not None and best[0] not in D and best[1] not in D:
                found = best
            for i in range(1, len(D)):
                p, q = D[i - 1], D[i]
                if g[i] >= ell and g[i] > g[i - 1] and g[i] > g[i + 1] and (p < a1) != (q < a1):
                    if q < a1:
                        p, q = q, p
                    if not (p == a0 or q == b0 or T[p - 1] != T[q - 1] or T[p - 1] in b"N$"):      # reveal.c:81-85 on the working text
                        continue
                    if found is None or g[i] > found[2] or (g[i] == found[2] and p < found[0]):
                        found = (p, q, g[i]); STATS["danger_hidden"] += 1
            if found is None and best is None:
                STATS["empty"] += 1
                continue
            if found is not None:
                qa, qb, ql = found
                anchors.append((ql, (qa, qb)))
                stack.append((a0, qa, b0, qb, live))
                stack.append((qa + ql, a1, qb + ql, b1, live))
                continue
            STATS["danger_fake"] += 1
        # not decided by the list: the child from its own text
        STATS["rebuilt"] += 1
        STATS["rebuilt_bases"] += (a1 - a0) + (b1 - b0)
        sub = [T[a0:a1].decode(), T[b0:b1].decode()]
        if depth > 40:
            raise RuntimeError("no progress")
        if (a1 - a0) + (b1 - b0) >= (nodes[0][1] - nodes[0][0]) + (nodes[1][1] - nodes[1][0]):
            got, _, _, _, _ = oracle_anchors(sub, minl)      # the whole input again: the literal recursion (what the library's regular path is)
        else:
            got = cascade(sub, minl, depth + 1)
        for (ql, (qa, qb)) in got:
            anchors.append((ql, (qa + a0, qb - (a1 - a0 + 1) + b0)))
    return sorted(anchors)


def main():
    from fuzz import make_case, mutate  # noqa: F401
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t_end = time.time() + budget
    bad = 0
    while time.time() < t_end:
        seqs, minl = make_case(rng)
        seqs = seqs[:2]
        if any(len(s) == 0 for s in seqs):
            continue
        # the text must be upper case for the oracle's assemble; N runs and repeats come from make_case
        want, _, _, _, _ = oracle_anchors(seqs, minl)
        got = cascade(seqs, minl)
        STATS["cases"] += 1
        STATS["total_bases"] += sum(len(s) for s in seqs)
        if got != want:
            bad += 1
            print("MISMATCH case", STATS["cases"], "minl", minl, "lens", [len(s) for s in seqs], "anchors", len(want), len(got))
            sw, sg = set(want), set(got)
            print("  only oracle:", sorted(sw - sg)[:5], " only cascade:", sorted(sg - sw)[:5])
            if bad >= 3:
                break
    print(STATS, "mismatches", bad)


if __name__ == "__main__":
    main()
