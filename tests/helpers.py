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
        if isinstance(f, str) and os.path.exists(f):
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
        if isinstance(f, str) and os.path.exists(f):
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
