#!/usr/bin/env python3
"""Pin the CPU restatement (liboracle) against the reference itself -- TEST INFRASTRUCTURE.

Runs in the build container only (needs /root/reference for the FASTA fixtures
and oracle/_ref/ built by `make -C oracle ref`).  For every fixture set:

  * SA: reference divsufsort == own prefix-doubling sorter (+ sufcheck)
  * LCP / SO: reference compute_lcp / build_SO == restatement
  * getmums, getmums_rem, getmultimums, getmultimems: reference == restatement
  * the full aligner recursion: a LIFO loop written here around the REFERENCE's
    getmums_rem/getmultimums + split + bubble_sort (D-label done by scatter
    through SAi as reveal.c:1005-1117) with the bench picker / linear
    graphalign, compared step by step (sub-index key, n, depth, nsamples,
    scan result, chosen match, child SA and LCP hashes) with ro_align's trace.
  * splitindex (reveal.c:1515-1748): ro_splitindex == label scatter + the
    reference's split + bubble_sort, two steps deep; extract (reveal.c:1386-1505):
    ro_extract == the reference's own extract() (ranks 1.. of SA -- the reference
    never writes SA[0] -- LCP, SAi, T), also with rc=1 interval remapping.
  * the known-answer vectors of SURVEY.md 8(c).
  * "aligner() itself": the reference's own recursion driver (reveal.c:731-1338, entered through index.align,
    interface.c:293-415) executed for real -- module oracle/_ref/reveallib[64].so = the reference's sources built as
    the CPython module they define (`make -C oracle refmod`, Python-2 API names mapped by oracle/refmod/py3_names.[ch])
    -- with the benchmark callbacks written as Python 3 functions of the reference's callback signatures.  Its
    per-callback trace (sub-index key, n, depth, nsamples, nodes, scan result, SA / LCP hashes of every sub-index it
    pops), the chosen matches and the final text must equal ro_align's AND the digests in tests/golden/vectors.json.

Exit code 0 = every check passed.  `python oracle/pin_oracle.py [--big]`.
"""
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_ctypes            # noqa: E402
import oracle_ctypes         # noqa: E402

REFTESTS = "/root/reference/tests"
GOLDEN = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def read_fasta(path, toupper=True):
    """reveal/utils.py:79-160 fasta_reader (defaults): rstrip, upper, drop '-'"""
    name, seq, out = None, [], []
    with open(path) as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(">"):
                if seq:
                    out.append((name, "".join(seq)))
                name, seq = line.replace(">", "").replace("\t", ""), []
            else:
                if toupper:
                    line = line.upper()
                seq.append(line.replace("-", ""))
    if seq:
        out.append((name, "".join(seq)))
    return out


def assemble(files, toupper=True):
    """reveal/utils.py:325-350 + interface.c:18-95: one sample per file, one
    '$'-terminated sequence per contig -> (T bytes, nsep, nodes)"""
    T, nsep, nodes = bytearray(), [], []
    for k, f in enumerate(files):
        if k > 0:
            nsep.append(len(T) - 1)
        for _, s in (read_fasta(f, toupper) if isinstance(f, str) and os.path.exists(f) else [(None, f)]):
            b = len(T)
            T += s.encode() + b"$"
            nodes.append((b, len(T) - 1))
    return bytes(T), nsep, nodes


def seqhash(vals):
    v = np.asarray(vals).astype(np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        x = v + (np.arange(1, len(v) + 1, dtype=np.uint64) * np.uint64(GOLDEN))
        x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
        x ^= x >> np.uint64(31)
        return int(x.sum(dtype=np.uint64))


def flat_mums(mums):
    out = []
    for l, n, spd in mums:
        out += [l, n]
        for so, pos in spd:
            out += [so, pos]
    return out


def bench_picker(mums, nsamples):
    best = None
    for m in mums:
        if m[1] != nsamples:
            continue
        mn = min(p for _, p in m[2])
        if best is None or m[0] > best[0][0] or (m[0] == best[0][0] and mn < best[1]):
            best = (m, mn)
    return best[0] if best else None


def linear_graphalign(nodes, mum):
    l, n, spd = mum
    lead, trail, match, touched = [], [], [], set()
    for _, sp in spd:
        hit = [q for q, (b, e) in enumerate(nodes) if b <= sp < e][0]
        b, e = nodes[hit]
        assert sp + l <= e
        touched.add(hit)
        if sp > b:
            lead.append((b, sp))
        if sp + l < e:
            trail.append((sp + l, e))
        match.append((sp, sp + l))
    rest = [iv for q, iv in enumerate(nodes) if q not in touched]
    return sorted(lead), sorted(trail), sorted(match), sorted(rest)


def ref_recursion(R, tbuf, SA, LCP, SAi, SO, nsep, nsamples, nodes, minl, minn, sa_t):
    """LIFO aligner loop (reveal.c:731-1338) around the reference's own scan,
    split and bubble_sort.  Yields trace dicts."""
    nsep_a = np.asarray(nsep, dtype=sa_t)

    def nsamp(iv):
        if nsamples > 2:
            return len({int(SO[b]) for b, _ in iv})
        s = set()
        for b, _ in iv:
            if b < nsep[0]:
                s.add(0)
            if b > nsep[0]:
                s.add(1)
        return len(s)
    main = R.view(tbuf, SA, LCP, nsep_a, nsamples, SAi=SAi, SO=SO)
    stack = [dict(SA=SA, LCP=LCP, depth=0, nsamples=nsamples, nodes=list(nodes))]
    while stack:
        ix = stack.pop()
        ri = R.view(tbuf, ix["SA"], ix["LCP"], nsep_a, nsamples, SAi=SAi, SO=SO, nT=len(SA), main=main)
        if nsamples > 2:
            mums = R.getmultimums(ri, minl, minn)
        else:
            mums = R.getmums_rem(ri, minl)
        rec = dict(key=min(b for b, _ in ix["nodes"]), n=len(ix["SA"]), depth=ix["depth"], nsamples=ix["nsamples"],
                   nnodes=len(ix["nodes"]), nmums=len(mums), h_sa=seqhash(ix["SA"]), h_lcp=seqhash(ix["LCP"]),
                   h_mums=seqhash(flat_mums(mums)) if mums else 0, picked=0, l=0, mn=0, sp_min=0)
        mum = bench_picker(mums, ix["nsamples"])
        if mum is None:
            yield rec
            continue
        lead, trail, match, rest = linear_graphalign(ix["nodes"], mum)
        rec.update(picked=1, l=mum[0], mn=mum[1], sp_min=min(p for _, p in mum[2]))
        yield rec
        n = len(ix["SA"])
        D = np.zeros(n, dtype=np.uint8)                     # reveal.c:1005-1117
        cnt = [0, 0, 0]
        for k, (ivs, lab) in enumerate(((lead, 1), (trail, 2), (rest, 4))):
            for b, e in ivs:
                D[SAi[b:e]] = lab
                cnt[k] += e - b
        for _, sp in mum[2]:
            D[SAi[sp:sp + mum[0]]] = 3
        kids = R.split(ri, D, *cnt)                          # reveal.c:1217
        for _, sp in mum[2]:                                 # reveal.c:1230-1234
            seg = tbuf[sp:sp + mum[0]]
            up = (seg >= 65) & (seg <= 90)
            seg[up] += 32
        if kids[0] is not None:                              # reveal.c:1250-1252
            R.bubble_sort(tbuf, kids[0][0], kids[0][1], SAi, match)
        d = ix["depth"] + 1
        if kids[2] is not None:                              # push par, lead, trail
            stack.append(dict(SA=kids[2][0], LCP=kids[2][1], depth=d, nsamples=nsamp(rest), nodes=rest))
        if kids[0] is not None:
            stack.append(dict(SA=kids[0][0], LCP=kids[0][1], depth=d, nsamples=nsamp(lead), nodes=lead))
        if kids[1] is not None:
            stack.append(dict(SA=kids[1][0], LCP=kids[1][1], depth=d, nsamples=nsamp(trail), nodes=trail))


def load_refmod(sa64):
    """the reference as the CPython module it defines (oracle/_ref/reveallib[64].so, `make -C oracle refmod`) or None"""
    import importlib.machinery
    import importlib.util
    name = "reveallib64" if sa64 else "reveallib"
    path = os.path.join(HERE, "_ref", name + ".so")
    if not os.path.exists(path):
        return None
    loader = importlib.machinery.ExtensionFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def run_real_aligner(mod, files, minl, minn):
    """index.align() of the reference itself (interface.c:293-415 -> aligner(), reveal.c:731-1338) with the benchmark
    callbacks in the reference's own callback signatures (reveal.c:839-999).  -> (trace records in pop order, final T)"""
    idx = mod.index()
    for f in files:
        if isinstance(f, str) and os.path.exists(f):
            idx.addsample(os.path.basename(f))
            for _, s in read_fasta(f):
                idx.addsequence(s)
        else:
            idx.addsample("lit")
            idx.addsequence(f)
    idx.construct()
    trace = []

    def mumpicker(mums, sub, precomputed=False, minlength=0):          # reveal.c:839-857: (multimums, idx, precomputed=, minlength=)
        nodes = sorted((int(b), int(e)) for b, e in sub.nodes)
        rec = dict(key=min(b for b, _ in nodes), n=int(sub.n), depth=int(sub.depth), nsamples=int(sub.nsamples), nnodes=len(nodes),
                   nmums=len(mums), h_sa=seqhash(sub.SA), h_lcp=seqhash(sub.LCP), h_mums=seqhash(flat_mums(mums)) if mums else 0,
                   picked=0, l=0, mn=0, sp_min=0, nodes=nodes)
        trace.append(rec)
        mum = bench_picker(mums, sub.nsamples)
        if mum is None:
            return ()                                                  # reveal.c:870-884
        rec.update(picked=1, l=int(mum[0]), mn=int(mum[1]), sp_min=min(int(p) for _, p in mum[2]))
        return (mum, [], [])                                           # (mum, skipmumsleft, skipmumsright), reveal.c:901

    def graphalign(sub, mum):                                          # reveal.c:939: (idx, mum) -> 7-tuple, :987
        lead, trail, match, rest = linear_graphalign(sorted((int(b), int(e)) for b, e in sub.nodes), mum)
        return (lead, trail, match, rest, None, None, None)

    idx.align(mumpicker, graphalign, threads=0, minl=minl, minn=minn)
    return trace, idx.T.encode("latin-1")


def pin_aligner(label, files, sa64, minl=20, minn=2, golden=None):
    print("[aligner() itself: %s]%s" % (label, " (64-bit)" if sa64 else ""))
    mod = load_refmod(sa64)
    if mod is None:
        check("reference module built (make -C oracle refmod)", False)
        return
    t0 = time.time()
    reft, finalT = run_real_aligner(mod, files, minl, minn)
    t_ref = time.time() - t0
    O = oracle_ctypes.Oracle(sa64)
    T, nsep, nodes = assemble(files)
    cons = O.construct(T, nsep, len(files))
    res = O.align_bench(cons, nodes, minl, minn, trace_cap=len(reft) + 16)
    tr = res["trace"]
    ok, bad = len(tr) == len(reft), None
    if ok:
        for k, r in enumerate(reft):          # same LIFO order on both sides: record by record
            for f in ("key", "n", "depth", "nsamples", "nnodes", "nmums", "picked", "l", "mn", "sp_min", "h_sa", "h_lcp", "h_mums"):
                if int(tr[k][f]) != int(r[f]) & (M64 if f.startswith("h_") else -1):
                    ok, bad = False, (k, f, int(tr[k][f]), r[f])
                    break
            if not ok:
                break
    check("aligner() trace == ro_align trace (every popped sub-index)", ok,
          "%d callbacks, %d anchors, aligner() %.1fs %s" % (len(reft), sum(r["picked"] for r in reft), t_ref, bad or ""))
    check("aligner() final T == ro_align final T", finalT == res["T"])
    if golden is not None:
        import hashlib
        import json
        g = golden["recursion"]
        anchors = sorted((r["l"], r["sp_min"], r["mn"]) for r in reft if r["picked"])
        key = sorted((r["depth"], r["key"], r["n"], r["nsamples"], r["nmums"], r["picked"], r["l"], r["sp_min"],
                      r["h_sa"] & M64, r["h_lcp"] & M64, r["h_mums"] & M64) for r in reft)
        check("aligner() trace == tests/golden/vectors.json sha_trace", hashlib.sha256(json.dumps(key).encode()).hexdigest() == g["sha_trace"], "%d steps" % g["steps"])
        check("aligner() anchors == vectors.json sha_anchors", hashlib.sha256(json.dumps(anchors).encode()).hexdigest() == g["sha_anchors"], "%d anchors" % g["anchors"])
        check("aligner() final T == vectors.json sha_finalT", hashlib.sha256(finalT).hexdigest() == g["sha_finalT"])


def csr_to_tuples(l, n, off, so, pos):
    return [(int(l[k]), int(n[k]), tuple((int(so[q]), int(pos[q])) for q in range(off[k], off[k + 1])))
            for k in range(len(l))]


def check(name, cond, info=""):
    print("  %-52s %s %s" % (name, "ok" if cond else "FAIL", info))
    if not cond:
        check.failed += 1


check.failed = 0


def pin_set(label, files, sa64, minl=20, minn=2, recursion=True, own_sa=True):
    print("[%s]%s" % (label, " (64-bit)" if sa64 else ""))
    R = ref_ctypes.Ref(sa64)
    O = oracle_ctypes.Oracle(sa64)
    T, nsep, nodes = assemble(files)
    n, ns = len(T), len(files)
    tb_r, tb_o = R.textbuf(T), O.textbuf(T)
    t0 = time.time()
    SA = R.divsufsort(tb_r[:n])
    t_dss = time.time() - t0
    if own_sa:
        t0 = time.time()
        SAo = O.suffix_array(tb_o, own=True)
        check("SA: own sorter == divsufsort", np.array_equal(SA, SAo), "n=%d ref %.2fs own %.2fs" % (n, t_dss, time.time() - t0))
    check("SA: sufcheck", O.sufcheck(tb_o, SA) == 0)
    SAi = R.inverse(SA)
    check("SAi", np.array_equal(SAi, O.inverse(SA)))
    LCP = R.compute_lcp(tb_r, SA, SAi)
    LCPo = O.compute_lcp(tb_o, SA, SAi)
    check("LCP: compute_lcp", np.array_equal(LCP, LCPo), "max=%d" % LCP.max())
    # closed form of SURVEY.md 7: min(plain lcp, distance to first '$'/'N')
    SO = None
    if ns > 2:
        SO = R.build_so(nsep, ns, n)
        check("SO: build_SO", np.array_equal(SO, O.build_so(nsep, ns, n)))
    ri = R.view(tb_r, SA, LCP, nsep, ns, SAi=SAi, SO=SO)
    if ns >= 2:
        for ml in sorted({1, minl}):
            ref = R.getmums(ri, ml)
            l, a, b = O.getmums(tb_o, SA, LCPo, nsep, ml)
            mine = [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]
            check("getmums(%d)" % ml, ref == mine, "%d mums" % len(ref))
        ref = R.getmums_rem(ri, minl)
        l, a, b = O.getmums(tb_o, SA, LCPo, nsep, minl, rem=True)
        mine = [(int(l[k]), 2, ((0, int(a[k])), (1, int(b[k])))) for k in range(len(l))]
        check("getmums_rem(%d)" % minl, ref == mine, "%d mums" % len(ref))
    if ns > 2:
        for ml, mn in ((minl, minn), (2, 2), (0, 2), (minl, ns)):
            ref = R.getmultimums(ri, ml, mn)
            mine = csr_to_tuples(*O.getmultimums(tb_o, SA, LCPo, SO, nsep, ns, ml, mn))
            check("getmultimums(%d,%d)" % (ml, mn), ref == mine, "%d" % len(ref))
            ref = R.getmultimems(ri, ml, mn)
            mine = csr_to_tuples(*O.getmultimums(tb_o, SA, LCPo, SO, nsep, ns, ml, mn, mems=True))
            check("getmultimems(%d,%d)" % (ml, mn), ref == mine, "%d" % len(ref))
    if recursion and ns >= 2:
        t0 = time.time()
        reft = list(ref_recursion(R, tb_r, SA.copy(), LCP.copy(), SAi.copy(), SO, nsep, ns, nodes, minl, minn, R.sa_t))
        t_ref = time.time() - t0
        cons = dict(tbuf=tb_o, SA=SA.copy(), SAi=SAi.copy(), LCP=LCPo.copy(), SO=SO,
                    nsep=np.asarray(nsep, dtype=O.sa_t), nsamples=ns)
        res = O.align_bench(cons, nodes, minl, minn, trace_cap=len(reft) + 16)
        tr = res["trace"]
        ok = len(tr) == len(reft)
        bad = None
        if ok:
            for k, r in enumerate(reft):
                t = tr[k]
                for f in ("key", "n", "depth", "nsamples", "nnodes", "nmums", "picked", "l", "mn", "sp_min", "h_sa", "h_lcp", "h_mums"):
                    if int(t[f]) != int(r[f]) & (M64 if f.startswith("h_") else -1):
                        ok, bad = False, (k, f, int(t[f]), r[f])
                        break
                if not ok:
                    break
        check("recursion trace (scan+label+split+bubble, LIFO)", ok,
              "%d steps, %d anchors, depth %d, ref-loop %.1fs %s" % (len(reft), res["stats"]["nsplits"], res["stats"]["maxdepth"], t_ref, bad or ""))
        check("final T (lower-case mask)", bytes(tb_r[:n]) == res["T"], "%d lower" % sum(1 for c in res["T"] if 97 <= c <= 122))



def pin_single_steps(label, files, sa64, minl=20, minn=2):
    """splitindex / extract of reveal.c:1386-1748, driven like the commented loop rem.py:580-609"""
    print("[single steps: %s]%s" % (label, " (64-bit)" if sa64 else ""))
    R = ref_ctypes.Ref(sa64)
    O = oracle_ctypes.Oracle(sa64)
    T, nsep, nodes = assemble(files)
    n, ns = len(T), len(files)
    tb_r, tb_o = R.textbuf(T), O.textbuf(T)
    SA = R.divsufsort(tb_r[:n]); SAi = R.inverse(SA); LCP = R.compute_lcp(tb_r, SA, SAi)
    SO = R.build_so(nsep, ns, n) if ns > 2 else None
    SAi_r, SAi_o = SAi.copy(), SAi.copy()
    main = R.view(tb_r, SA, LCP, nsep, ns, SAi=SAi_r, SO=SO)
    work = [(SA, LCP, list(nodes), ns, 0)]
    steps = 0
    first_match = None
    while work and steps < 7:
        sa, lcp, nd, nsub, depth = work.pop(0)
        ri = R.view(tb_r, sa, lcp, nsep, ns, SAi=SAi_r, SO=SO, nT=n, main=main)
        mums = R.getmultimums(ri, minl, minn) if ns > 2 else R.getmums_rem(ri, minl)
        mum = bench_picker(mums, nsub)
        if mum is None:
            continue
        lead, trail, match, rest = linear_graphalign(nd, mum)
        if first_match is None:
            first_match = match
        # reference side: label (scatter, reveal.c:1548-1640) + split + bubble_sort
        D = np.zeros(len(sa), dtype=np.uint8)
        cnt = [0, 0, 0]
        for b, e in lead:
            D[SAi_r[b:e]] = 1; cnt[0] += e - b
        for b, e in trail:
            D[SAi_r[b:e]] = 2; cnt[1] += e - b
        for b, e in match:
            D[SAi_r[b:e]] = 3
            seg = tb_r[b:e]; up = (seg >= 65) & (seg <= 90); seg[up] += 32
        for b, e in rest:
            D[SAi_r[b:e]] = 4; cnt[2] += e - b
        kids_r = R.split(ri, D, *cnt)
        if kids_r[0] is not None:
            R.bubble_sort(tb_r, kids_r[0][0], kids_r[0][1], SAi_r, match)
        kids_o = O.splitindex(tb_o, sa, lcp, SAi_o, SO, nsep, ns, lead, trail, match, rest)
        ok = all((a is None) == (b is None) and (a is None or (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))) for a, b in zip(kids_r, kids_o))
        # (SAi entries of dropped ranks keep stale values on both sides)
        ok = ok and np.array_equal(SAi_r, SAi_o) and bytes(tb_r) == bytes(tb_o)
        check("splitindex step %d (depth %d, n=%d, l=%d)" % (steps, depth, len(sa), mum[0]), ok)
        for k, ivs in enumerate((lead, trail, rest)):
            if kids_o[k] is not None:
                work.append((kids_o[k][0], kids_o[k][1], ivs, kids_o[k][2], depth + 1))
        steps += 1
    # extract on the untouched main index: the first match of the run above, then a second, disjoint set on the result
    for rc in (0, 1):
        tb_r, tb_o = R.textbuf(T), O.textbuf(T)
        SAi_r, SAi_o = SAi.copy(), SAi.copy()
        ivs = list(first_match) if first_match else [(1, 3)]
        if rc == 1:       # hand in query-side intervals in reverse-complement coordinates (reveal.c:1411-1427 maps them back)
            ivs = [(b, e) if b <= nsep[0] else (nsep[0] + n - e, nsep[0] + n - b) for b, e in ivs]
        sa_r, lcp_r, iv_r = R.extract(tb_r, SA, LCP, SAi_r, nsep, ivs, rc=rc, nT=n)
        sa_o, lcp_o, iv_o = O.extract(tb_o, SA, LCP, SAi_o, nsep, ivs, rc=rc, nT=n)
        ok = len(sa_r) == len(sa_o) and np.array_equal(sa_r[1:], sa_o[1:]) and np.array_equal(lcp_r, lcp_o) and sa_o[0] == SA[0]
        ok = ok and np.array_equal(SAi_r, SAi_o) and bytes(tb_r) == bytes(tb_o)
        check("extract (rc=%d, %d intervals, %d -> %d ranks)" % (rc, len(ivs), n, len(sa_o)), ok)
        if rc == 0:
            check("extract rc=0 leaves the intervals alone", iv_o == [(int(b), int(e)) for b, e in ivs])
        else:
            check("extract rc=1 remaps query intervals", iv_o == [(int(b), int(e)) for b, e in first_match] if first_match else True)
        if rc == 0 and first_match:
            b0 = first_match[0][0]
            second = [(max(b0 - 40, nodes[0][0]), max(b0 - 25, nodes[0][0] + 1))]
            if second[0][0] < second[0][1] <= b0:
                sa_r2, lcp_r2, _ = R.extract(tb_r, sa_o, lcp_o, SAi_r, nsep, second, nT=n)
                sa_o2, lcp_o2, _ = O.extract(tb_o, sa_o, lcp_o, SAi_o, nsep, second, nT=n)
                ok = np.array_equal(sa_r2[1:], sa_o2[1:]) and np.array_equal(lcp_r2, lcp_o2) and np.array_equal(SAi_r, SAi_o) and bytes(tb_r) == bytes(tb_o)
                check("extract again on the result (%d ranks)" % len(sa_o2), ok)


def known_answers():
    print("[known answers, SURVEY.md 8(c)]")
    for sa64 in (False, True):
        O = oracle_ctypes.Oracle(sa64)
        T, nsep, nodes = assemble(["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"])
        check("T/nsep/nodes", T == b"ACTTGCTAGCTAGTCAG$ACTAGCTAGCTAGTGAG$" and nsep == [17] and nodes == [(0, 17), (18, 35)])
        c = O.construct(T, nsep, 2)
        check("SA", list(c["SA"]) == [35, 17, 18, 0, 33, 15, 21, 7, 25, 11, 29, 14, 19, 5, 23, 9, 27, 1, 34, 16, 32, 4, 22, 8, 26, 12, 30, 20, 6, 24, 10, 28, 13, 31, 3, 2])
        check("LCP", list(c["LCP"]) == [0, 0, 0, 3, 1, 2, 2, 6, 7, 2, 3, 0, 1, 8, 9, 4, 5, 2, 0, 1, 1, 1, 10, 5, 6, 1, 2, 0, 7, 8, 3, 4, 1, 1, 2, 1])
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 1)
        check("getmums(1)", list(zip(l, a, b)) == [(3, 0, 18), (10, 4, 22), (2, 3, 31)])
        T, nsep, nodes = assemble(["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"])
        c = O.construct(T, nsep, 3)
        check("3 samples n/nsep", len(T) == 54 and nsep == [17, 35])
        mm = csr_to_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, 3, 2, 2))
        check("getmultimums(2,2)", mm == [(9, 2, ((0, 0), (2, 36))), (3, 3, ((1, 18), (0, 0), (2, 36))), (10, 2, ((0, 4), (1, 22))),
                                          (7, 2, ((2, 46), (0, 10))), (4, 3, ((2, 46), (0, 10), (1, 28))), (2, 3, ((1, 31), (0, 3), (2, 39)))])
        me = csr_to_tuples(*O.getmultimums(c["tbuf"], c["SA"], c["LCP"], c["SO"], nsep, 3, 4, 3, mems=True))
        check("getmultimems(4,3)", me == [(5, 3, ((0, 4), (1, 22), (2, 40), (0, 8), (1, 26))), (4, 3, ((2, 46), (0, 10), (1, 28)))])
    # complement table (interface.c:136-145) vs ro_revcomp
    import ctypes
    R = ref_ctypes.Ref(False)
    tab = (ctypes.c_ubyte * 128).in_dll(R.lib, "comp_tab")
    O = oracle_ctypes.Oracle(False)
    buf = np.arange(128, dtype=np.uint8)
    O.revcomp(buf)
    check("comp_tab", list(buf[::-1]) == list(tab))


def main():
    big = "--big" in sys.argv
    f = lambda *names: [os.path.join(REFTESTS, x + ".fa") for x in names]
    known_answers()
    pin_set("t1+t2 (degenerate 1-bp contigs)", f("t1", "t2"), False, minl=1)
    pin_set("1a+1b (config 1)", f("1a", "1b"), False)
    pin_set("1a+1b (config 1)", f("1a", "1b"), True)
    pin_set("1a+1b+1c (3-way)", f("1a", "1b", "1c"), False)
    pin_set("1a+1b+1c+1d+1e (5-way, multi-contig)", f("1a", "1b", "1c", "1d", "1e"), False)
    pin_set("1e+1b (multi-contig)", f("1e", "1b"), False)
    pin_set("d1+d2 (50k N run)", f("d1", "d2"), False)
    pin_set("1a+1brc", f("1a", "1brc"), False)
    pin_set("1a+1a (identical)", f("1a", "1a"), False)
    pin_set("2a+2b", f("2a", "2b"), False, own_sa=big)
    pin_single_steps("1a+1b", f("1a", "1b"), False)
    pin_single_steps("1a+1b", f("1a", "1b"), True)
    pin_single_steps("1a+1b+1c", f("1a", "1b", "1c"), False)
    pin_single_steps("1e+1b (multi-contig)", f("1e", "1b"), False)
    gold = {}
    try:
        import json
        with open(os.path.join(os.path.dirname(HERE), "tests", "golden", "vectors.json")) as fh:
            gold = json.load(fh)["sets"]
    except Exception:
        pass
    pin_aligner("known answers, 2 samples", ["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"], False, minl=1, golden=gold.get("known2"))
    pin_aligner("t1+t2", f("t1", "t2"), False, minl=1, golden=gold.get("t1t2"))
    pin_aligner("1a+1b (config 1)", f("1a", "1b"), False, golden=gold.get("1a1b"))
    pin_aligner("1a+1b (config 1)", f("1a", "1b"), True, golden=gold.get("1a1b_64"))
    pin_aligner("1a+1b+1c (3-way)", f("1a", "1b", "1c"), False, golden=gold.get("1a1b1c"))
    pin_aligner("1a+1b+1c+1d+1e (5-way, multi-contig)", f("1a", "1b", "1c", "1d", "1e"), False, golden=gold.get("5way"))
    pin_aligner("1e+1b (multi-contig)", f("1e", "1b"), False, golden=gold.get("1e1b"))
    pin_aligner("d1+d2 (50k N run)", f("d1", "d2"), False, golden=gold.get("d1d2"))
    if big:
        pin_set("3a+3b", f("3a", "3b"), False, own_sa=False)
        pin_set("1a+1b+1c (3-way)", f("1a", "1b", "1c"), True)
    print("FAILED: %d" % check.failed if check.failed else "ALL PINNED")
    return 1 if check.failed else 0


if __name__ == "__main__":
    sys.exit(main())
