"""Size-independent properties of a finished recursion (numpy; used by tests/ and bench.py at sizes the CPU oracle
does not finish in seconds).  Not a compute path: it only looks at what the HIP path returned.

What align() must leave behind whatever the input (reveallib/reveal.c:731-1338 with the benchmark callbacks of
SURVEY.md 8(d); the first two are the array form of the reference's only round-trip test, tests/test_reveal.py:150-159):

  text_spells_input      upper-casing the final text gives the input text back
  lower_mask_is_anchors  exactly the members of the anchors are lower case (reveal.c:1230-1234), no two members overlap
  anchors_exact          every member of every anchor spells the same l bases (checked for ALL anchors, not a sample)
  anchors_minl           every anchor has l >= minl
  anchors_full           every anchor has one member in each of its samples' ... (members strictly ascending = distinct samples)
  anchors_collinear      ordered by their position in the first sample, anchors that share samples are ordered the same way
                         in every other sample (the linear interval model never crosses two anchors) -- asserted for anchors
                         present in every sample
"""
import numpy as np


def _upper(t):
    return np.where((t >= 97) & (t <= 122), t - 32, t).astype(np.uint8)


def recursion_properties(T0, T1, anchors, nsep, minl, chunk=1 << 25, collinear=True):
    """T0 / T1: uint8 arrays (or bytes) of the text before / after align; anchors = (l, off, pos) as align_builtin
    returns them; nsep = separator positions (sample boundaries).  -> dict of booleans + counts"""
    T0 = np.frombuffer(T0, dtype=np.uint8) if isinstance(T0, (bytes, bytearray)) else np.asarray(T0, dtype=np.uint8)
    T1 = np.frombuffer(T1, dtype=np.uint8) if isinstance(T1, (bytes, bytearray)) else np.asarray(T1, dtype=np.uint8)
    l, off, pos = anchors
    l = np.asarray(l, dtype=np.int64); off = np.asarray(off, dtype=np.int64); pos = np.asarray(pos, dtype=np.int64)
    n = len(T0)
    out = {"anchors": int(len(l)), "anchored_bp": int(l.sum()), "n": int(n)}
    out["text_spells_input"] = bool(len(T1) == n and np.array_equal(_upper(T1), T0))
    cnt = np.diff(off)
    ll = np.repeat(l, cnt)
    d = np.zeros(n + 1, dtype=np.int32)
    np.add.at(d, pos, 1)
    np.add.at(d, pos + ll, -1)
    cover = np.cumsum(d[:-1], dtype=np.int32)
    lower = (T1 >= 97) & (T1 <= 122)
    out["lower_mask_is_anchors"] = bool((cover.max() if n else 0) <= 1 and np.array_equal(lower, cover == 1))
    del cover, lower, d
    out["anchors_minl"] = bool(len(l) == 0 or l.min() >= max(minl, 1))
    # members of an anchor come sorted by position: strictly ascending samples = one member per sample
    seps = np.asarray(nsep, dtype=np.int64)
    so = np.searchsorted(seps, pos, side="left")
    first = np.zeros(len(pos), dtype=bool)
    first[off[:-1]] = True
    asc = np.ones(len(pos), dtype=bool)
    asc[1:] = (so[1:] > so[:-1]) | first[1:]
    out["anchors_full"] = bool(asc.all() and (cnt >= 2).all())
    # exact matches: every member against the first member of its anchor, all bases, in chunks
    ok = True
    first_pos = np.repeat(pos[off[:-1]], cnt)
    other = ~first
    op, fp, ol = pos[other], first_pos[other], ll[other]
    start = 0
    csum = np.cumsum(ol)
    while start < len(op) and ok:
        base = csum[start - 1] if start else 0
        stop = int(np.searchsorted(csum, base + chunk, side="right"))
        stop = max(stop, start + 1)
        lens = ol[start:stop]
        tot = int(lens.sum())
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        ia = np.repeat(op[start:stop], lens) + within
        ib = np.repeat(fp[start:stop], lens) + within
        ok = bool(np.array_equal(T0[ia], T0[ib]))
        start = stop
    out["anchors_exact"] = ok
    # collinearity over the anchors present in every sample
    ns = len(seps) + 1
    full = cnt == ns
    if full.any() and collinear:      # (samples of several sequences in different orders are collinear per pair of sequences only: not asserted)
        m = pos[(off[:-1][full][:, None] + np.arange(ns)[None, :]).ravel()].reshape(-1, ns)
        order = np.argsort(m[:, 0], kind="stable")
        out["anchors_collinear"] = bool((np.diff(m[order], axis=0) > 0).all())
    else:
        out["anchors_collinear"] = True
    out["all"] = all(v for k, v in out.items() if isinstance(v, bool))
    return out


# ---- digests of a whole result (what tests/golden/fullsize.json holds, written by oracle/gen_fullsize_golden.py) ----
def array_digest(a):
    """sha256 of an array's bytes as they lie in memory (SA: int32 / int64, LCP: int32 / uint32, text: uint8)"""
    import hashlib
    a = np.ascontiguousarray(a)
    h = hashlib.sha256()
    mv = memoryview(a).cast("B")
    step = 1 << 28
    for s in range(0, len(mv), step):
        h.update(mv[s:s + step])
    return h.hexdigest()


def anchor_stream(l, off, pos):
    """canonical form of an anchor set: anchors ordered by their first member's position (members never overlap, so that is a total
    order), each written as l, member count, member positions -- one int64 stream"""
    l = np.asarray(l, dtype=np.int64); off = np.asarray(off, dtype=np.int64); pos = np.asarray(pos, dtype=np.int64)
    na = len(l)
    if na == 0:
        return np.zeros(0, dtype=np.int64)
    cnt = np.diff(off)
    order = np.argsort(pos[off[:-1]], kind="stable")
    c = cnt[order]
    start = np.cumsum(c + 2) - (c + 2)
    out = np.empty(int((c + 2).sum()), dtype=np.int64)
    out[start] = l[order]
    out[start + 1] = c
    within = np.arange(int(c.sum()), dtype=np.int64) - np.repeat(np.cumsum(c) - c, c)
    out[np.repeat(start + 2, c) + within] = pos[np.repeat(off[:-1][order], c) + within]
    return out


def anchor_digest(l, off, pos):
    return array_digest(anchor_stream(l, off, pos))


def golden_record(L, genomes, seed, indelfrac=0.0, minl=20, minn=2, path=None, snp=0.01, repeats=0.0, nruns=0):
    """the CPU path's digests for this synthetic configuration (tests/golden/fullsize.json, written by oracle/gen_fullsize_golden.py in the
    build container from the reference's divsufsort + the restated recursion), or None when the file holds none"""
    import json
    import os
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")
    if not os.path.exists(path):
        return None
    for name, r in json.load(open(path)).items():
        if (r["L"], r["genomes"], r["seed"], float(r["indelfrac"]), r["minl"], r["minn"], float(r.get("snp", 0.01)), float(r.get("repeats", 0.0)), int(r.get("nruns", 0))) == \
           (L, genomes, seed, float(indelfrac), minl, minn, float(snp), float(repeats), int(nruns)):
            return dict(r, name=name)
    return None


def compare_with_golden(rec, anchors=None, T_final=None, SA=None, LCP=None):
    """-> dict: which of the given results equal the CPU path's digests (only the ones passed in are looked at)"""
    out = {"golden": rec["name"]}
    if SA is not None:
        out["SA"] = array_digest(SA) == rec["sha_SA"]
    if LCP is not None:
        out["LCP"] = array_digest(LCP) == rec["sha_LCP"]
    if anchors is not None:
        l, off, pos = anchors
        out["anchor_count"] = int(len(l)) == rec["anchors"]
        out["anchors"] = anchor_digest(l, off, pos) == rec["sha_anchors"]
    if T_final is not None:
        out["final_text"] = array_digest(np.asarray(T_final, dtype=np.uint8)) == rec["sha_finalT"]
    out["all"] = all(v for k, v in out.items() if isinstance(v, bool))
    return out
