"""Seeded synthetic genomes for the benchmark configurations.

Base genome: i.i.d. uniform over ACGT, length L, numpy PCG64(seed).  Variant k:
copy of the base with exactly floor(L*snp) distinct uniformly chosen positions
substituted by a uniformly chosen *different* base, PCG64(seed + k).  One contig
per genome.  The same bytes feed the GPU path and the CPU baseline.

`indelfrac > 0` switches the variants to the mutation model of the reference's own
simulator (utils/simulate.py:17-77 `mut`): ceil(rate * L) distinct uniformly chosen
positions, visited in ascending order; an event is an indel with probability
`indelfrac` (half insertions, half deletions, lengths zipf(1.7) capped at 2000; an
inserted base is drawn from A, C, G only and a substituted base from the first two
of the three other letters -- simulate.py:41, 52 draw with randint(0, 3) / randint(0, 2),
whose upper bound is exclusive), otherwise a substitution; positions swallowed by
an earlier deletion are skipped (simulate.py:35-36).  The draws come from PCG64
(seed + k) in a fixed vectorised order, not from the simulator's global Mersenne
state: the distribution is the simulator's, the stream is ours.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def base_codes(L, seed=42):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 4, size=L, dtype=np.uint8)


def _distinct_positions(rng, L, count):
    pos = np.empty(0, dtype=np.int64)
    while len(pos) < count:
        pos = np.unique(np.concatenate([pos, rng.integers(0, L, size=int(count * 1.1) + 16)]))
    return rng.permutation(pos)[:count]


def variant_codes(base, k, seed=42, snp=0.01):
    L = len(base)
    rng = np.random.Generator(np.random.PCG64(seed + k))
    nsub = int(L * snp)
    pos = _distinct_positions(rng, L, nsub)
    out = base.copy()
    out[pos] = (out[pos] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
    return out


def variant_codes_indel(base, k, seed=42, rate=0.01, indelfrac=0.2, zipfd=1.7, maxindellength=2000):
    """the simulator's `mut` (utils/simulate.py:17-77) on 2-bit codes, vectorised.  -> codes of the variant"""
    L = len(base)
    rng = np.random.Generator(np.random.PCG64(seed + k))
    npos = int(np.ceil(rate * L))
    pos = np.sort(_distinct_positions(rng, L, npos))
    indel = rng.random(npos) < indelfrac
    ins = indel & (rng.random(npos) < 0.5)
    dele = indel & ~ins
    length = np.minimum(rng.zipf(zipfd, size=npos), maxindellength).astype(np.int64)
    alt = rng.integers(0, 2, size=npos, dtype=np.uint8)          # which of the first two other letters (simulate.py:52)
    # positions inside an earlier, not itself skipped, deletion are never visited (simulate.py:35-36)
    skipped = np.zeros(npos, dtype=bool)
    while True:      # an event's fate only depends on the events in front of it: every pass settles a longer prefix
        reach = np.maximum.accumulate(np.where(dele & ~skipped, pos + length, 0))
        now = np.zeros(npos, dtype=bool)
        now[1:] = pos[1:] < reach[:-1]
        if np.array_equal(now, skipped):
            break
        skipped = now
    live = ~skipped
    out = base.copy()
    s = live & ~indel
    b = base[pos[s]]
    # "ACGT".replace(b, "")[alt]: the alt-th of the remaining letters in ACGT order
    out[pos[s]] = np.where(alt[s] >= b, alt[s] + 1, alt[s]).astype(np.uint8)
    d = live & dele
    dd = np.zeros(L + 1, dtype=np.int32)
    np.add.at(dd, pos[d], 1)
    np.add.at(dd, np.minimum(pos[d] + length[d], L), -1)
    keep = (np.cumsum(dd[:L]) == 0)
    del dd
    i = live & ins
    before = np.zeros(L, dtype=np.int64)
    before[pos[i]] = length[i]
    new_index = np.cumsum(keep, dtype=np.int64) - 1 + np.cumsum(before)
    total = int(keep.sum()) + int(length[i].sum())
    res = np.full(total, 255, dtype=np.uint8)
    res[new_index[keep]] = out[keep]
    del new_index, before
    hole = res == 255
    res[hole] = rng.integers(0, 3, size=int(hole.sum()), dtype=np.uint8)      # A, C, G (simulate.py:41, 47)
    return res


def member(base, k, seed=42, snp=0.01, indelfrac=0.0):
    """member k of the family genomes(L, count, seed) makes, from the base's codes (k = 0: the base itself)"""
    if k == 0:
        return _ACGT[base].tobytes()
    if indelfrac > 0:
        return _ACGT[variant_codes_indel(base, k, seed, snp, indelfrac)].tobytes()
    return _ACGT[variant_codes(base, k, seed, snp)].tobytes()


def genomes(L, count, seed=42, snp=0.01, indelfrac=0.0):
    """-> list of `count` byte strings: the base and count-1 variants of it."""
    base = base_codes(L, seed)
    out = [_ACGT[base].tobytes()]
    for k in range(1, count):
        if indelfrac > 0:
            out.append(_ACGT[variant_codes_indel(base, k, seed, snp, indelfrac)].tobytes())
        else:
            out.append(_ACGT[variant_codes(base, k, seed, snp)].tobytes())
    return out
