#!/usr/bin/env python3
"""Generate tests/golden/vectors.json from the REFERENCE itself -- TEST INFRASTRUCTURE.

Runs only in the build container: needs /root/reference (FASTA fixtures, already
copied as data to tests/golden/*.fa.gz) and oracle/_ref/ (the reference's own C
built unmodified, `make -C oracle ref`).  For every fixture set it records what
the reference computes -- array digests, scan results, and the full recursion
under the benchmark callbacks (LIFO loop around the reference's getmums_rem /
getmultimums + split + bubble_sort, see oracle/pin_oracle.py:ref_recursion) --
so that tests on the GPU box (no /root/reference there) can check the oracle and
the HIP path against reference output.

    python oracle/gen_golden.py        # rewrites tests/golden/vectors.json
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import pin_oracle as P          # noqa: E402
import ref_ctypes               # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    """a fixture is either a tests/golden/<name>.fa.gz copy of a reference test file or a literal sequence"""
    p = os.path.join(GOLD, name + ".fa.gz")
    if os.path.exists(p):
        tmp = os.path.join("/tmp", "golden_" + name + ".fa")
        with gzip.open(p, "rt") as f, open(tmp, "w") as g:
            g.write(f.read())
        return tmp
    return name


def one(label, inputs, minl, minn=2, sa64=False):
    R = ref_ctypes.Ref(sa64)
    files = [load(x) for x in inputs]
    T, nsep, nodes = P.assemble(files)
    n, ns = len(T), len(files)
    tb = R.textbuf(T)
    SA = R.divsufsort(tb[:n])
    SAi = R.inverse(SA)
    LCP = R.compute_lcp(tb, SA, SAi)
    SO = R.build_so(nsep, ns, n) if ns > 2 else None
    ri = R.view(tb, SA, LCP, nsep, ns, SAi=SAi, SO=SO)
    rec = dict(inputs=inputs, sa64=sa64, minl=minl, minn=minn, n=n, nsep=nsep, nodes=nodes,
               sha_T=hashlib.sha256(T).hexdigest(), sha_SA=sha(SA.astype(np.int64)), sha_LCP=sha(LCP.astype(np.int64)),
               maxlcp=int(LCP.max()))
    if n <= 64:
        rec.update(T=T.decode(), SA=[int(x) for x in SA], LCP=[int(x) for x in LCP])
    mums = R.getmums(ri, minl)
    rec["getmums"] = dict(minl=minl, count=len(mums), sha=hashlib.sha256(json.dumps(mums).encode()).hexdigest(), head=mums[:8])
    if ns > 2:
        mm = R.getmultimums(ri, minl, minn)
        rec["getmultimums"] = dict(count=len(mm), sha=hashlib.sha256(json.dumps(mm).encode()).hexdigest(), head=mm[:4])
    # extract() (reveal.c:1386-1505) of the longest full match, on fresh copies; the reference never writes the new SA[0]
    full = [m for m in (R.getmultimums(ri, minl, minn) if ns > 2 else R.getmums_rem(ri, minl)) if m[1] == ns]
    if full and n > 64:
        pick = P.bench_picker(full, ns)
        ivs = [(int(p), int(p) + int(pick[0])) for _, p in pick[2]]
        tb2, SAi2 = R.textbuf(T), SAi.copy()
        xsa, xlcp, _ = R.extract(tb2, SA, LCP, SAi2, nsep, ivs, nT=n)
        rec["extract"] = dict(intervals=ivs, n=len(xsa), sha_SA1=sha(xsa[1:].astype(np.int64)), sha_LCP=sha(xlcp.astype(np.int64)),
                              sha_T=hashlib.sha256(bytes(tb2[:n])).hexdigest())
    trace = list(P.ref_recursion(R, tb, SA.copy(), LCP.copy(), SAi.copy(), SO, nsep, ns, nodes, minl, minn, R.sa_t))
    anchors = sorted((r["l"], r["sp_min"], r["mn"]) for r in trace if r["picked"])
    key = sorted((r["depth"], r["key"], r["n"], r["nsamples"], r["nmums"], r["picked"], r["l"], r["sp_min"],
                  r["h_sa"] & P.M64, r["h_lcp"] & P.M64, r["h_mums"] & P.M64) for r in trace)
    finalT = bytes(tb[:n])
    rec["recursion"] = dict(steps=len(trace), anchors=len(anchors), anchored_bp=sum(a[0] for a in anchors),
                            maxdepth=max(r["depth"] for r in trace),
                            sha_anchors=hashlib.sha256(json.dumps(anchors).encode()).hexdigest(),
                            sha_trace=hashlib.sha256(json.dumps(key).encode()).hexdigest(),
                            sha_finalT=hashlib.sha256(finalT).hexdigest(),
                            lowercase=sum(1 for c in finalT if 97 <= c <= 122), anchors_head=anchors[:6])
    print("%-28s n=%-8d mums=%-6d steps=%-6d anchors=%d" % (label, n, len(mums), len(trace), len(anchors)))
    return rec


def seeded(label, inputs, minl, minn=2, sa64=False):
    """the reference's REAL aligner() (module oracle/_ref/reveallib[64].so, `make -C oracle refmod`) under a picker that seeds
    its children (non-empty skipmums: reveal.c:802, 830-837, 1157, 1180) -- the callbacks are reveal_amd/rem.py's, the same
    objects the GPU test hands to reveal_amd's index.align"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from reveal_amd import rem
    mod = P.load_refmod(sa64)
    idx = mod.index()
    for x in inputs:
        f = load(x)
        if os.path.exists(f):
            idx.addsample(os.path.basename(f))
            for _, s in P.read_fasta(f):
                idx.addsequence(s)
        else:
            idx.addsample("lit"); idx.addsequence(f)
    idx.construct()
    pick, galign, trace = H.traced_callbacks(rem.seeding_mumpicker, rem.linear_graphalign)
    idx.align(pick, galign, threads=0, minl=minl, minn=minn)
    d = H.callback_trace_digest(trace)
    d.update(inputs=inputs, minl=minl, minn=minn, sa64=sa64, sha_finalT=hashlib.sha256(idx.T.encode("latin-1")).hexdigest())
    print("%-28s seeded: calls=%d precomputed=%d anchors=%d" % (label, d["calls"], d["precomputed_calls"], d["anchors"]))
    return d


def main():
    sets = {
        "known2": (["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"], 1, 2),
        "known3": (["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG", "ACTTGCTAGGTAGTCAG"], 2, 2),
        "t1t2": (["t1", "t2"], 1, 2),
        "1a1b": (["1a", "1b"], 20, 2),
        "1a1b1c": (["1a", "1b", "1c"], 20, 2),
        "5way": (["1a", "1b", "1c", "1d", "1e"], 20, 2),
        "1e1b": (["1e", "1b"], 20, 2),
        "d1d2": (["d1", "d2"], 20, 2),
        "1a1brc": (["1a", "1brc"], 20, 2),
        # round 5: the rest of the reference's fixtures (tests/1f.fa, e2.fa, and its two Mbp-scale pairs 2a+2b, 3a+3b:
        # the only inputs with real repeat structure the reference ships)
        "1a1f": (["1a", "1f"], 20, 2),
        "1ae2": (["1a", "e2"], 20, 2),
        "2a2b": (["2a", "2b"], 20, 2),
        "3a3b": (["3a", "3b"], 20, 2),
    }
    only = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    if only:                                  # add / refresh some sets, keep the rest of the file
        with open(os.path.join(GOLD, "vectors.json")) as f:
            out = json.load(f)
        for label in only[0]:
            inputs, minl, minn = sets[label]
            out["sets"][label] = one(label, inputs, minl, minn)
        with open(os.path.join(GOLD, "vectors.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print("updated tests/golden/vectors.json:", ",".join(only[0]))
        return
    out = {"_generator": "oracle/gen_golden.py (reference C built unmodified into oracle/_ref)", "sets": {}}
    for label, (inputs, minl, minn) in sets.items():
        out["sets"][label] = one(label, inputs, minl, minn)
    out["sets"]["1a1b_64"] = one("1a1b_64", ["1a", "1b"], 20, 2, sa64=True)
    if P.load_refmod(False) is not None:      # aligner() itself with seeded children (skipmums)
        out["seeded"] = {"1a1b": seeded("1a1b", ["1a", "1b"], 20), "1a1b1c": seeded("1a1b1c", ["1a", "1b", "1c"], 20),
                         "5way": seeded("5way", ["1a", "1b", "1c", "1d", "1e"], 20), "1a1b_64": seeded("1a1b_64", ["1a", "1b"], 20, sa64=True)}
    with open(os.path.join(GOLD, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote tests/golden/vectors.json")


if __name__ == "__main__":
    main()
