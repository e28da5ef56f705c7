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

`repeats > 0` (round 5: the unfriendly classes of the bench's class table and of the full-size
digests) overlays the BASE with what real genomes have and i.i.d. text has not: interspersed
copies of a dozen element families (1-6 kb consensus, every copy 0-5 % diverged from it, a
third truncated; a fifth of the copies exact: ties far beyond the first key, beyond the text
round's 4 KB) covering `repeats` of the length, tandem arrays (unit 2-60 bp x 10-300) and --
`nruns` -- runs of N (10 bp - 50 kb, the same places in every member: assembly gaps).  All of
it from PCG64(seed + 1000003), so a family is a function of (L, seed, repeats, nruns).
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


def overlay_repeats(base, seed=42, repeats=0.02, tandem=None):
    """interspersed repeat families and tandem arrays written over the base's codes (in place) -> base
    tandem: fraction of the base in tandem arrays (None: one array per 2 Mbp, the default family's ~0.4 %)"""
    L = len(base)
    rng = np.random.Generator(np.random.PCG64(seed + 1000003))
    nfam = 12
    cons = [rng.integers(0, 4, size=int(rng.integers(1000, 6001)), dtype=np.uint8) for _ in range(nfam)]
    weight = 1.0 / np.arange(1, nfam + 1)
    weight /= weight.sum()
    budget = int(L * repeats)
    while budget > 0:
        f = int(rng.choice(nfam, p=weight))
        c = cons[f]
        if rng.random() < 0.33 and len(c) > 400:                 # truncated copy
            a = int(rng.integers(0, len(c) - 200)); b = int(rng.integers(a + 200, len(c) + 1))
            c = c[a:b]
        c = c.copy()
        if rng.random() >= 0.2:                                  # (a fifth of the copies stay exact)
            nsub = int(len(c) * rng.random() * 0.05)
            if nsub:
                q = rng.integers(0, len(c), size=nsub)
                c[q] = (c[q] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
        if len(c) >= L:
            break
        at = int(rng.integers(0, L - len(c)))
        base[at:at + len(c)] = c
        budget -= len(c)
    narr, tbudget = max(2, L // 2_000_000), None
    if tandem is not None:
        narr, tbudget = 1 << 60, int(L * tandem)
    k = 0
    while k < narr and (tbudget is None or tbudget > 0):         # tandem arrays
        k += 1
        unit = rng.integers(0, 4, size=int(rng.integers(2 if tandem is None else 1, 61)), dtype=np.uint8)
        arr = np.tile(unit, int(rng.integers(10, 301)))
        if len(arr) >= L:
            continue
        at = int(rng.integers(0, L - len(arr)))
        base[at:at + len(arr)] = arr
        if tbudget is not None:
            tbudget -= len(arr)
    return base


def n_runs(L, seed=42, nruns=0):
    """-> sorted list of (begin, end) of the runs of N (log-uniform lengths 10 .. 50 000, clipped to the text)"""
    if nruns <= 0:
        return []
    rng = np.random.Generator(np.random.PCG64(seed + 2000003))
    out = []
    for _ in range(nruns):
        ln = int(min(10 ** rng.uniform(1.0, 4.7), max(1, L // 8)))
        at = int(rng.integers(0, max(1, L - ln)))
        out.append((at, min(L, at + ln)))
    return sorted(out)


def _spell(codes, runs):
    out = _ACGT[codes]
    for b, e in runs:
        out[b:min(e, len(out))] = 78      # 'N'
    return out.tobytes()


def family(L, count, seed=42, snp=0.01, indelfrac=0.0, repeats=0.0, nruns=0, tandem=None):
    """genomes() with the unfriendly knobs: `repeats` (fraction of the base covered by interspersed repeat copies; tandem arrays
    come with it) and `nruns` (runs of N, the same text positions in every member).  repeats = 0 and nruns = 0: genomes()."""
    base = base_codes(L, seed)
    if repeats > 0 or tandem:
        overlay_repeats(base, seed, repeats, tandem)
    runs = n_runs(L, seed, nruns)
    out = [_spell(base, runs)]
    for k in range(1, count):
        v = variant_codes_indel(base, k, seed, snp, indelfrac) if indelfrac > 0 else variant_codes(base, k, seed, snp)
        out.append(_spell(v, runs))
    return out


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
