"""shared test helpers (CPU side).  The oracle is the checker, never the thing tested."""
import functools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from reveal_amd import rem, synth          # noqa: E402  (pure python, no GPU needed)
from oracle import oracle_ctypes           # noqa: E402


def fa(*names):
    return [os.path.join(GOLD, n + ".fa.gz") for n in names]


def assemble(inputs, toupper=True):
    """reference text-assembly protocol (utils.py:325-350, interface.c:18-95) on the host:
    inputs = FASTA paths (one sample per file, one sequence per contig) or raw
    sequences (one sample each).  -> (T bytes, nsep, nodes)"""
    T, nsep, nodes = bytearray(), [], []
    for k, f in enumerate(inputs):
        if k > 0:
            nsep.append(len(T) - 1)
        if isinstance(f, (list, tuple)):                       # one sample given as its sequences (a multi-contig FASTA without the file)
            seqs = [c.decode() if isinstance(c, (bytes, bytearray)) else c for c in f]
        elif isinstance(f, str) and os.path.exists(f):
            seqs = [s for _, s in rem.fasta_reader(f, toupper=toupper)]
        else:
            seqs = [f.decode() if isinstance(f, (bytes, bytearray)) else f]
        for s in seqs:
            b = len(T)
            T += s.encode() + b"$"
            nodes.append((b, len(T) - 1))
    return bytes(T), nsep, nodes


@functools.lru_cache(maxsize=None)
def oracle(sa64=False):
    return oracle_ctypes.Oracle(sa64)


def feed(idx, inputs, toupper=True):
    """same protocol, into an index object"""
    for k, f in enumerate(inputs):
        if isinstance(f, (list, tuple)):
            idx.addsample("s%d" % k)
            for c in f:
                idx.addsequence(c)
        elif isinstance(f, str) and os.path.exists(f):
            rem.read_fasta(f, idx, toupper=toupper)
        else:
            idx.addsample("s%d" % k)
            idx.addsequence(f)
    return idx


def csr_tuples(l, n, off, so, pos):
    return [(int(l[k]), int(n[k]), tuple((int(so[q]), int(pos[q])) for q in range(off[k], off[k + 1]))) for k in range(len(l))]


# ---- golden vectors produced by the reference itself (oracle/gen_golden.py) ----
import hashlib   # noqa: E402
import json      # noqa: E402

M64 = (1 << 64) - 1


def golden():
    with open(os.path.join(GOLD, "vectors.json")) as f:
        return json.load(f)["sets"]


def golden_seeded():
    """digests of the reference's real aligner() under rem.seeding_mumpicker (oracle/gen_golden.py `seeded`)"""
    with open(os.path.join(GOLD, "vectors.json")) as f:
        return json.load(f).get("seeded", {})


def golden_inputs(rec):
    return [os.path.join(GOLD, x + ".fa.gz") if os.path.exists(os.path.join(GOLD, x + ".fa.gz")) else x for x in rec["inputs"]]


def sha_arr(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a).astype(np.int64)).tobytes()).hexdigest()


def sha_json(obj):
    return hashlib.sha256(json.dumps(obj).encode()).hexdigest()


def trace_digests(tr):
    """(sha_trace, sha_anchors, anchors, anchored_bp) of a structured trace array, as gen_golden.py computes them"""
    key = sorted((int(r["depth"]), int(r["key"]), int(r["n"]), int(r["nsamples"]), int(r["nmums"]), int(r["picked"]), int(r["l"]),
                  int(r["sp_min"]), int(r["h_sa"]) & M64, int(r["h_lcp"]) & M64, int(r["h_mums"]) & M64) for r in tr)
    anchors = sorted((int(r["l"]), int(r["sp_min"]), int(r["mn"])) for r in tr if r["picked"])
    return sha_json(key), sha_json(anchors), len(anchors), sum(a[0] for a in anchors)


# ---- callback tracing: the same wrapper drives the reference's own module (oracle/gen_golden.py) and reveal_amd's index
GOLDEN64 = 0x9E3779B97F4A7C15


def seqhash(vals):
    """order-sensitive 64-bit digest of an integer sequence (same function as oracle/pin_oracle.py seqhash / ro_hash_step)"""
    v = np.asarray(vals).astype(np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        x = v + (np.arange(1, len(v) + 1, dtype=np.uint64) * np.uint64(GOLDEN64))
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


def traced_callbacks(mumpicker, graphalign, arrays=True):
    """-> (mumpicker', graphalign', trace): one record per mumpicker call with what the callback could see of its sub-index"""
    trace = []

    def pick(mums, sub, precomputed=False, minlength=0):
        nodes = sorted((int(b), int(e)) for b, e in sub.nodes)
        rec = dict(depth=int(sub.depth), key=min(b for b, _ in nodes), n=int(sub.n), nsamples=int(sub.nsamples), nnodes=len(nodes),
                   precomputed=1 if precomputed else 0, nmums=len(mums), h_mums=seqhash(flat_mums(mums)) if len(mums) else 0,
                   h_sa=seqhash(sub.SA) if arrays else 0, h_lcp=seqhash(sub.LCP) if arrays else 0, picked=0, l=0, mn=0, sp_min=0)
        trace.append(rec)
        r = mumpicker(mums, sub, precomputed=precomputed, minlength=minlength)
        if isinstance(r, tuple) and len(r) == 3:
            m = r[0]
            rec.update(picked=1, l=int(m[0]), mn=int(m[1]), sp_min=min(int(p) for _, p in m[2]))
        return r
    return pick, graphalign, trace


def callback_trace_digest(trace):
    key = sorted((r["depth"], r["key"], r["n"], r["nsamples"], r["nnodes"], r["precomputed"], r["nmums"], r["picked"], r["l"], r["mn"], r["sp_min"],
                  r["h_sa"] & M64, r["h_lcp"] & M64, r["h_mums"] & M64) for r in trace)
    anchors = sorted((r["l"], r["sp_min"], r["mn"]) for r in trace if r["picked"])
    return dict(calls=len(trace), precomputed_calls=sum(r["precomputed"] for r in trace), anchors=len(anchors),
                sha_trace=sha_json(key), sha_anchors=sha_json(anchors))
