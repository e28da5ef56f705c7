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
