"""Seeded synthetic genomes for the benchmark configurations.

Base genome: i.i.d. uniform over ACGT, length L, numpy PCG64(seed).  Variant k:
copy of the base with exactly floor(L*snp) distinct uniformly chosen positions
substituted by a uniformly chosen *different* base, PCG64(seed + k).  One contig
per genome.  The same bytes feed the GPU path and the CPU baseline.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def base_codes(L, seed=42):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 4, size=L, dtype=np.uint8)


def variant_codes(base, k, seed=42, snp=0.01):
    L = len(base)
    rng = np.random.Generator(np.random.PCG64(seed + k))
    nsub = int(L * snp)
    pos = np.empty(0, dtype=np.int64)
    while len(pos) < nsub:
        pos = np.unique(np.concatenate([pos, rng.integers(0, L, size=int(nsub * 1.1) + 16)]))
    pos = rng.permutation(pos)[:nsub]
    out = base.copy()
    out[pos] = (out[pos] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
    return out


def genomes(L, count, seed=42, snp=0.01):
    """-> list of `count` byte strings: the base and count-1 variants of it."""
    base = base_codes(L, seed)
    out = [_ACGT[base].tobytes()]
    for k in range(1, count):
        out.append(_ACGT[variant_codes(base, k, seed, snp)].tobytes())
    return out
