"""construct() and getmums() on the GPU, bit-exact against the CPU oracle
(reveallib/interface.c:160-291, reveallib/reveal.c:55-116)."""
import numpy as np
import pytest

from helpers import assemble, fa, feed, oracle, synth

pytestmark = pytest.mark.gpu

SETS = {
    "known": ["ACTTGCTAGCTAGTCAG", "ACTAGCTAGCTAGTGAG"],
    "t1t2": fa("t1", "t2"),
    "1a1b": fa("1a", "1b"),
    "1a1b1c": fa("1a", "1b", "1c"),
    "1e1b": fa("1e", "1b"),
    "d1d2": fa("d1", "d2"),
    "1a1a": fa("1a", "1a"),
    "5way": fa("1a", "1b", "1c", "1d", "1e"),
}


def mod(sa64):
    from reveal_amd import reveallib, reveallib64
    return reveallib64 if sa64 else reveallib


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("name", list(SETS))
def test_construct_matches_oracle(name, sa64):
    T, nsep, nodes = assemble(SETS[name])
    O = oracle(sa64)
    c = O.construct(T, nsep, len(SETS[name]))
    idx = feed(mod(sa64).index(), SETS[name])
    assert idx.n == len(T) and idx.nsep == nsep and sorted(idx.nodes) == nodes
    idx.construct()
    assert idx.T.encode("latin-1") == T
    assert np.array_equal(idx.array("SA"), c["SA"])
    assert np.array_equal(idx.array("SAi"), c["SAi"])
    assert np.array_equal(idx.array("LCP"), c["LCP"])
    if len(SETS[name]) > 2:
        assert np.array_equal(idx.array("SO"), c["SO"])
    for minl in (1, 20):
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, minl)
        assert idx.getmums(minl) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def test_known_answer_vectors():
    """SURVEY.md 8(c) / reveal/tests/test_reveal.py:37 input"""
    idx = feed(mod(False).index(), SETS["known"])
    idx.construct()
    assert idx.T == "ACTTGCTAGCTAGTCAG$ACTAGCTAGCTAGTGAG$"
    assert idx.SA == [35, 17, 18, 0, 33, 15, 21, 7, 25, 11, 29, 14, 19, 5, 23, 9, 27, 1, 34, 16, 32, 4, 22, 8, 26, 12, 30, 20, 6, 24, 10, 28, 13, 31, 3, 2]
    assert idx.LCP == [0, 0, 0, 3, 1, 2, 2, 6, 7, 2, 3, 0, 1, 8, 9, 4, 5, 2, 0, 1, 1, 1, 10, 5, 6, 1, 2, 0, 7, 8, 3, 4, 1, 1, 2, 1]
    assert idx.getmums(1) == [(3, (0, 18), 0), (10, (4, 22), 0), (2, (3, 31), 0)]


@pytest.mark.parametrize("L,cnt", [(1000, 2), (50000, 2), (300000, 3)])
def test_construct_synthetic(L, cnt):
    seqs = [g.decode() for g in synth.genomes(L, cnt)]
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, cnt)
    idx = feed(mod(False).index(), seqs)
    idx.construct()
    assert np.array_equal(idx.array("SA"), c["SA"])
    assert np.array_equal(idx.array("LCP"), c["LCP"])
    l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 20)
    assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def test_getmums_dense_hits_overflow():
    """diverged samples (10 % SNP): tens of thousands of short MUMs, more per 512-rank tile than its own slots hold and more in
    all than the first overflow buffer -- the scan is repeated with a larger one (and must not read what it could not store)"""
    seqs = [g.decode() for g in synth.genomes(1500000, 2, seed=5, snp=0.1)]
    T, nsep, nodes = assemble(seqs)
    O = oracle(False)
    c = O.construct(T, nsep, 2)
    idx = feed(mod(False).index(), seqs)
    idx.construct()
    for minl in (12, 11):
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, minl)
        got = idx.getmums(minl)
        assert len(got) == len(l) > 30000
        assert got == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def _repeat_cases():
    import random
    rng = random.Random(11)
    unit = "".join(rng.choice("ACGT") for _ in range(700))
    return {
        # twelve near-copies: every first-key group has twelve members (the ranked-by-text path for 9..64 members)
        "twelve_samples": [g.decode() for g in synth.genomes(20000, 12)],
        # a period-2 and a period-3 run of 6000 symbols: first-key groups of thousands, ties far beyond the text-compare cap
        # (radix path for large groups, then doubling rounds)
        "periodic": ["AC" * 3000 + unit, "ACG" * 2000 + unit[::-1]],
        # one repeated unit inside each sample (groups of ~20) plus a homopolymer
        "tandem": [(unit[:300] * 20) + "A" * 5000, (unit[:300] * 20) + "C" * 100],
        # 70 identical short contigs in one sample: groups of 70 (> 64) that only the '$' positions tell apart
        "many_contigs": ["$".join([unit[:50]] * 70), unit[:50]],
    }


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("name", ["twelve_samples", "periodic", "tandem", "many_contigs"])
def test_construct_repeats(name, sa64):
    """group sizes and match lengths that take every branch of the SA refinement (registers / ranked / radix, text round / doubling)"""
    seqs = _repeat_cases()[name]
    if name == "many_contigs":      # contigs of one sample: one addsequence per contig
        contigs = seqs[0].split("$")
        inputs = [contigs, [seqs[1]]]
        from reveal_amd import reveallib, reveallib64
        idx = (reveallib64 if sa64 else reveallib).index()
        for k, cs in enumerate(inputs):
            idx.addsample("s%d" % k)
            for cseq in cs:
                idx.addsequence(cseq)
        T = ("$".join(contigs) + "$" + seqs[1] + "$").encode()
        nsep = [len("$".join(contigs))]
        nsamples = 2
    else:
        T, nsep, nodes = assemble(seqs)
        idx = feed(mod(sa64).index(), seqs)
        nsamples = len(seqs)
    O = oracle(sa64)
    c = O.construct(T, nsep, nsamples)
    idx.construct()
    assert idx.T.encode("latin-1") == T
    assert np.array_equal(idx.array("SA"), c["SA"])
    assert np.array_equal(idx.array("LCP"), c["LCP"])


def test_construct_rc_and_files(tmp_path, monkeypatch):
    """construct(rc=1) remap (interface.c:168-175, reveal.c:98-100) and the sa=/lcp=/cache= files"""
    monkeypatch.chdir(tmp_path)
    T, nsep, nodes = assemble(fa("1a", "1brc"))
    O = oracle(False)
    tb = O.textbuf(T)
    O.revcomp(tb[nsep[0]:len(T)])
    SA = O.suffix_array(tb); SAi = O.inverse(SA); LCP = O.compute_lcp(tb, SA, SAi)
    l, a, b = O.getmums(tb, SA, LCP, nsep, 20, rc=1, nT=len(T))
    idx = feed(mod(False).index(cache=1), fa("1a", "1brc"))
    idx.construct(rc=1)
    assert np.array_equal(idx.array("SA"), SA) and np.array_equal(idx.array("LCP"), LCP)
    assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 1) for k in range(len(l))]
    assert (tmp_path / ".reveal.sa").stat().st_size == 4 * len(T)
    # read the cached arrays back instead of computing
    idx2 = feed(mod(False).index(sa=str(tmp_path / ".reveal.sa"), lcp=str(tmp_path / ".reveal.lcp")), fa("1a", "1brc"))
    idx2.construct(rc=1)
    assert np.array_equal(idx2.array("SA"), SA) and np.array_equal(idx2.array("LCP"), LCP)


def test_errors():
    m = mod(False)
    idx = m.index()
    with pytest.raises(m.error):
        idx.construct()                       # "No text to index."
    with pytest.raises(m.error):
        idx.addsample(3)
    idx.addsample("a"); idx.addsequence("ACGT")
    with pytest.raises(TypeError):
        idx.SA                                # "Index not yet constructed."


def test_text_growth_after_construct_unconstructs():
    """addsample / addsequence after construct(): the arrays in HBM describe the old text, so every getter and scan
    refuses until the next construct() (never an out-of-bounds device read)"""
    m = mod(False)
    idx = feed(m.index(), SETS["known"])
    idx.construct()
    n0 = idx.n
    assert len(idx.SA) == n0
    idx.addsample("late")
    idx.addsequence("ACTTGCTAGGTAGTCAG")
    assert idx.n == n0 + 18
    with pytest.raises(TypeError):
        idx.SA
    with pytest.raises(TypeError):
        idx.LCP
    with pytest.raises((TypeError, m.error)):
        idx.getmums(1)
    assert idx.T == "ACTTGCTAGCTAGTCAG$ACTAGCTAGCTAGTGAG$ACTTGCTAGGTAGTCAG$"      # the host text, all of it
    idx.construct()
    T, nsep, nodes = assemble(SETS["known"] + ["ACTTGCTAGGTAGTCAG"])
    c = oracle(False).construct(T, nsep, 3)
    assert np.array_equal(idx.array("SA"), c["SA"]) and np.array_equal(idx.array("LCP"), c["LCP"])


def test_corrupt_sa_file_is_an_error(tmp_path, monkeypatch):
    """sa= files are range- and permutation-checked on the device before anything scatters through them (interface.c:224-232
    reads them unchecked)"""
    monkeypatch.chdir(tmp_path)
    m = mod(False)
    idx = feed(m.index(cache=1), fa("1a", "1b"))
    idx.construct()
    sa = np.fromfile(tmp_path / ".reveal.sa", dtype=np.int32)
    for kind in ("range", "dup", "short"):
        bad = sa.copy()
        if kind == "range":
            bad[len(bad) // 2] = len(bad) + 12345
        elif kind == "dup":
            bad[7] = bad[8]
        else:
            bad = bad[:-5]
        bad.tofile(tmp_path / "bad.sa")
        idx2 = feed(m.index(sa=str(tmp_path / "bad.sa")), fa("1a", "1b"))
        with pytest.raises(m.error):
            idx2.construct()
    # an index larger than the 32-bit rank range is refused whatever the source of SA (ranks travel as u32 inside the library)
    good = feed(m.index(sa=str(tmp_path / ".reveal.sa"), lcp=str(tmp_path / ".reveal.lcp")), fa("1a", "1b"))
    good.construct()
    assert np.array_equal(good.array("SA"), sa)


@pytest.mark.parametrize("mode", ["0", "2"])
@pytest.mark.parametrize("name", ["1a1b", "5way", "d1d2", "1a1a"])
def test_text_round_variants(monkeypatch, name, mode):
    """the other two work divisions of the SA build's text round (first thread orders its whole group / every member ranks
    itself; the default mixes them) and the fused LCP / BWT they emit: same SA, LCP and scan results"""
    monkeypatch.setenv("RV_TEXT_MODE", mode)
    T, nsep, nodes = assemble(SETS[name])
    O = oracle(False)
    c = O.construct(T, nsep, len(SETS[name]))
    idx = feed(mod(False).index(), SETS[name])
    idx.construct()
    assert np.array_equal(idx.array("SA"), c["SA"]) and np.array_equal(idx.array("LCP"), c["LCP"])
    l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 20)
    assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))]


def test_fused_lcp_equals_the_separate_pass(monkeypatch):
    """LCP / BWT written by the SA build itself (head LCP from the keys, member LCP from the text round) against the PHI / PLCP
    pass (RV_NO_FUSED_LCP), on inputs with N runs, '$' inside the first-key window and lower case"""
    seqs = ["ACGTNNNNNNNNNNACGTACGTTTGACCANNACGT" * 40 + "ACGTAC", "ACGTNNNNNNNNNNACGTACGTTTGACCANNACGA" * 40 + "ACGTAC", "ACG", "A", "acgtacgtACGTacgtNNacgt" * 30]
    seqs += [g.decode() for g in synth.genomes(30000, 3, seed=9)]
    m = mod(False)

    def build():
        idx = m.index()
        for k, s_ in enumerate(seqs):
            idx.addsample("s%d" % k)
            idx.addsequence(s_)
        idx.construct()
        return idx.array("SA"), idx.array("LCP"), idx.getmultimums(5, 2), idx.maxlcp
    fused = build()
    monkeypatch.setenv("RV_NO_FUSED_LCP", "1")
    plain = build()
    assert np.array_equal(fused[0], plain[0]) and np.array_equal(fused[1], plain[1])
    assert fused[2] == plain[2] and fused[3] == plain[3]
    T, nsep, nodes = assemble(seqs, toupper=False)
    c = oracle(False).construct(T, nsep, len(seqs))
    assert np.array_equal(fused[0], c["SA"]) and np.array_equal(fused[1], c["LCP"])


@pytest.mark.parametrize("sa64", [False, True])
def test_maxlcp_counts_the_group_heads(sa64):
    """The fused SA/LCP build takes the index' largest LCP from two places: members of a first-key group (text round) and the
    group heads (common prefix of two keys).  Unrelated short inputs have no group with two members at all, so the maximum
    is a head's value; it bounds the cut windows of bubble_sort (reveal.c:666-727), so it must not come out too small."""
    rng = np.random.default_rng(7)
    for L in (40, 300, 3000):
        seqs = ["".join("ACGT"[x] for x in rng.integers(0, 4, L)) for _ in range(2)]
        idx = feed(mod(sa64).index(), seqs)
        idx.construct()
        lcp = idx.array("LCP")
        assert idx.maxlcp == int(lcp.max()), (L, idx.maxlcp, int(lcp.max()))
    # every group cut short by a '$' / 'N' inside its key
    idx = feed(mod(sa64).index(), ["ACGTNACGTNACGTN", "ACGTNACGT"])
    idx.construct()
    assert idx.maxlcp == int(idx.array("LCP").max())


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("collapse", [True, False])
def test_twins_leave_before_the_sort(monkeypatch, collapse, sa64):
    """two samples with the diagonal hint: the suffixes of the second sample that carry their homologue's first key are not
    sorted (k_tw_count / k_init_keys / k_heads_publish_tc) -- SA, LCP, the largest LCP and the matches equal the oracle's on
    related samples of equal and different length, identical samples, N runs and lower case next to the tile and word borders,
    a second sample longer than twice the first, and inputs of a few bases"""
    if not collapse:
        monkeypatch.setenv("RV_NO_TWIN_COLLAPSE", "1")
    rng = np.random.default_rng(23)

    def rnd(L):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, L))

    def snp(s, rate):
        a = list(s)
        for p in np.nonzero(rng.random(len(a)) < rate)[0]:
            a[p] = "ACGT"[rng.integers(0, 4)]
        return "".join(a)
    base = rnd(70000)
    cases = [
        [base, snp(base, 0.01)],
        [base, base],
        [base[:40000], snp(base, 0.01)],                                   # the second sample longer
        [base, snp(base[:25000], 0.02)],                                   # ... shorter
        [base[:9000], snp(base[:9000], 0.01) + rnd(30000)],                # ... longer than twice the first
        [base[:1023] + "N" * 3 + base[1026:5000], snp(base[:5000], 0.005)],
        [base[:63] + "n" + base[64:3000], base[:3000]],
        [base[:2048], base[1:2049]],                                       # related on another diagonal only
        ["ACGTACGTACGTACGTACGTACGT" * 50, "ACGTACGTACGTACGTACGTACGT" * 50],
        ["ACGTA", "ACGTA"], ["A", "A"], ["ACGTACGTACGTACGTAC", "ACGTACGTACGTACGTAC"],
    ]
    for seqs in cases:
        T, nsep, nodes = assemble(seqs, toupper=False)
        O = oracle(sa64)
        c = O.construct(T, nsep, 2)
        idx = mod(sa64).index()
        for k, s_ in enumerate(seqs):
            idx.addsample("s%d" % k)
            idx.addsequence(s_)
        idx.construct()
        tag = (len(seqs[0]), len(seqs[1]))
        assert np.array_equal(idx.array("SA"), c["SA"]), tag
        assert np.array_equal(idx.array("LCP"), c["LCP"]), tag
        assert idx.maxlcp == int(c["LCP"].max()), tag
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 20)
        assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))], tag
        if collapse and len(seqs[0]) == 70000 and len(seqs[1]) == 70000:
            assert idx.sa_stats()["sorted_elems"] < idx.n, idx.sa_stats()      # some twins did leave


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("mode", ["auto", "forced", "off"])
def test_piecewise_diagonals(monkeypatch, mode, sa64):
    """two samples that left their fixed diagonal (indels): the hint and the twins' leaving follow a table of diagonals made from seeds
    (k_seed_sample / k_seed_pairs / k_dtab_fill / k_diag_bits_tab; RV_DIAG_TABLE 1 = always, 0 = never) -- SA, LCP, the largest LCP
    and the matches equal the oracle's on the reference simulator's mutation model, on indels next to tile and word borders, on
    insertions longer than a tile, on a second sample that starts with an insertion / ends with a deletion, on identical samples
    (nothing to find), unrelated samples (no seeds at all) and a rearranged second sample (seeds that contradict each other)"""
    if mode != "auto":
        monkeypatch.setenv("RV_DIAG_TABLE", "1" if mode == "forced" else "0")
    rng = np.random.default_rng(29)

    def rnd(L):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, L))
    base = rnd(90000)
    sim = [g.decode() for g in synth.genomes(120000, 2, seed=5, indelfrac=0.2)]
    sim_dense = [g.decode() for g in synth.genomes(60000, 2, seed=6, snp=0.03, indelfrac=0.5)]
    cases = [
        sim, sim_dense,
        [base, base[:63] + base[64:20000]],                                 # a deletion at the first word border
        [base[:30000], base[:1024] + "G" + base[1024:30000]],               # an insertion at a key-tile border
        [base[:30000], base[:5000] + rnd(3000) + base[5000:30000]],         # an insertion longer than a tile (and than the seeds' reach)
        [base[:30000], rnd(700) + base[:30000]],                            # the second sample starts with an insertion
        [base[:30000], base[:29000]],                                       # ... ends early
        [base[:40000], base[20000:40000] + base[:20000]],                   # rearranged: two diagonals, both far from the fixed one
        [base[:20000], rnd(20000)],                                         # unrelated
        [base[:20000], base[:20000]],
        [base[:20000], base[:10000] + base[10001:20000]],
        ["ACGTACGTACGTTTACGTACGT" * 300, "ACGTACGTACGTTACGTACGT" * 300],    # repeats: every seed is ambiguous
        [base[:5000], base[1:5000]], ["ACGTA", "ACGA"],
    ]
    used = 0
    for seqs in cases:
        T, nsep, nodes = assemble(seqs, toupper=False)
        O = oracle(sa64)
        c = O.construct(T, nsep, 2)
        idx = mod(sa64).index()
        for k, s_ in enumerate(seqs):
            idx.addsample("s%d" % k)
            idx.addsequence(s_)
        idx.construct()
        tag = (len(seqs[0]), len(seqs[1]), idx.sa_stats())
        assert np.array_equal(idx.array("SA"), c["SA"]), tag
        assert np.array_equal(idx.array("LCP"), c["LCP"]), tag
        assert idx.maxlcp == int(c["LCP"].max()), tag
        l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 20)
        assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))], tag
        used += idx.sa_stats()["diag_table"]
        if seqs is sim and mode != "off":
            st = idx.sa_stats()
            assert st["diag_table"] == 1 and st["sorted_elems"] < 0.8 * idx.n, st      # the table was used and most twins left
    assert (used > 0) == (mode != "off")


@pytest.mark.parametrize("sa64", [False, True])
def test_first_key_layouts_for_every_alphabet_size(sa64):
    """the first key packs its symbols in fields of g digits (rv_build_sa picks g per alphabet: three base-5 digits in seven bits for DNA, one
    symbol per field for 2, 4 or many letters, five base-3 digits in eight bits ...): SA, LCP and the largest LCP equal the oracle's for alphabets
    of 1 to 88 letters, related and unrelated samples, with and without 'N' -- and the layout the library reports is one of the expected ones"""
    rng = np.random.default_rng(99)
    letters = [chr(c) for c in range(65, 91) if chr(c) != "N"] + [chr(c) for c in range(97, 123)] + list("0123456789") + list("!#%&()*+,-./:;<=>?@[]^_{|}~")
    seen_layouts = set()
    for nsym in (1, 2, 3, 4, 5, 6, 7, 12, 20, 26, 60, 88):
        alpha = letters[:nsym]
        for L, related in ((3000, True), (700, False)):
            a = "".join(alpha[x] for x in rng.integers(0, nsym, L))
            if related:
                b = list(a)
                for p in np.nonzero(rng.random(L) < 0.02)[0]:
                    b[p] = alpha[rng.integers(0, nsym)]
                b = "".join(b)
                if nsym in (4, 12):
                    b = b[:500] + "N" * 7 + b[507:]
            else:
                b = "".join(alpha[x] for x in rng.integers(0, nsym, L + 13))
            seqs = [a, b]
            T, nsep, nodes = assemble(seqs, toupper=False)
            c = oracle(sa64).construct(T, nsep, 2)
            idx = feed(mod(sa64).index(), seqs)
            idx.construct()
            tag = (nsym, L, related)
            assert np.array_equal(idx.array("SA"), c["SA"]), tag
            assert np.array_equal(idx.array("LCP"), c["LCP"]), tag
            assert idx.maxlcp == int(c["LCP"].max()), tag
            st = idx.sa_stats()
            seen_layouts.add((st["sigma"], st["k0"], st["bits"]))
    assert len(seen_layouts) >= 8, seen_layouts


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("count,L,seed", [(5, 40000, 1), (6, 30011, 2), (9, 20000, 3), (10, 60000, 4), (12, 25000, 5), (15, 9000, 6), (16, 8000, 7), (17, 8000, 8)])
def test_text_round_of_many_samples(count, L, seed, sa64):
    """groups of one homologue per sample rank themselves from the words their members leave in LDS (k_round_text3: the order among the
    homologues of a base, entries staged on either side of the workgroup): group sizes 5 to 17 (the hint knows fifteen samples; seventeen
    members: past the staged border entries), groups across every workgroup border, variants that share a substitution (equal agreement
    with the base: the text decides), a sample that repeats another, a repeat inside the base -- SA and LCP against the oracle"""
    rng = np.random.default_rng(100 + seed)
    gs = [bytearray(g) for g in synth.genomes(L, count, seed=seed, snp=0.01)]
    # substitutions shared by two or three variants (they leave the base at the same place, on the same side)
    for _ in range(L // 200):
        p = int(rng.integers(L)); ch = b"ACGT"[int(rng.integers(4))]
        for k in rng.choice(np.arange(1, count), size=min(count - 1, int(rng.integers(2, 4))), replace=False):
            gs[int(k)][p] = ch
    if count > 5:
        gs[count - 1] = bytearray(gs[2])                     # a sample twice
    if L >= 20000:
        gs[0][5000:5600] = gs[0][1000:1600]                  # a repeat inside the base (the variants keep their own text there)
    seqs = [bytes(g).decode() for g in gs]
    T, nsep, nodes = assemble(seqs)
    idx = feed(mod(sa64).index(), seqs)
    c = oracle(sa64).construct(T, nsep, count)
    idx.construct()
    assert np.array_equal(idx.array("SA"), c["SA"])
    assert np.array_equal(idx.array("LCP"), c["LCP"])


def test_reset_reuses_the_handle():
    """rv_reset (not in the reference): a handle forgets its text and samples, keeps its allocations, and indexes the next input as a
    fresh handle would -- incl. a text large enough to live in page-locked host memory (one DMA copy into HBM) between two small ones"""
    import hashlib
    from reveal_amd import check, reveallib
    from helpers import assemble, feed, oracle, synth

    def run(idx, seqs, minl=20):
        feed(idx, seqs)
        idx.construct()
        sa, lcp = idx.array("SA").copy(), idx.array("LCP").copy()
        r = idx.align_builtin(minl, 2)
        l, off, pos = r["anchors"]
        return sa, lcp, sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l))), hashlib.sha256(idx.T.encode("latin-1")).hexdigest()

    small = [g.decode() for g in synth.genomes(200000, 2, seed=3)]
    three = [g.decode() for g in synth.genomes(90000, 3, seed=4)]
    idx = reveallib.index()
    a1 = run(idx, small)
    idx.reset()
    assert idx.n == 0 and idx.nsamples == 0 and len(idx.nodes) == 0 and idx.samples == []
    with pytest.raises(Exception):
        idx.getmums(20)                                   # not constructed
    b1 = run(idx, three)
    idx.reset(reserve=2 * 5_000_001)
    big = synth.genomes(5_000_000, 2, seed=42)            # 10 MB of text: page-locked
    feed(idx, [g.decode() for g in big])
    idx.construct()
    rec = check.golden_record(5_000_000, 2, 42)
    g = check.compare_with_golden(rec, SA=idx.array("SA"), LCP=idx.array("LCP"))
    assert g["all"], g
    res = idx.align_builtin(20, 2)
    g = check.compare_with_golden(rec, anchors=res["anchors"], T_final=idx.array("T"))
    assert g["all"], g
    idx.reset()
    a2 = run(idx, small)
    for x, y in zip(a1, a2):
        assert np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y
    fresh = run(reveallib.index(), three)
    for x, y in zip(b1, fresh):
        assert np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y
    T, nsep, nodes = assemble(small)
    c = oracle(False).construct(T, nsep, 2)
    assert np.array_equal(a2[0], c["SA"]) and np.array_equal(a2[1], c["LCP"])


@pytest.mark.parametrize("sa64", [False, True])
@pytest.mark.parametrize("mode", ["default", "no_far", "no_list", "no_jump", "tab", "far_table", "no_slow"])
def test_ties_beyond_the_text_round(monkeypatch, mode, sa64):
    """what the first key and the text round leave tied (agreement beyond 4 KB): near-identical and identical pairs are read off the diagonal's
    marks (k_far_twins), repeats are ordered by the doubling rounds and get LCP / BWT from the text afterwards (k_lcp_list) instead of a rebuild
    of the whole index; every switch's old path, and the piecewise diagonals, give the same arrays -- SA, LCP, the largest LCP, the matches and
    the recursion's anchors equal the oracle's"""
    env = {"no_far": "RV_NO_FAR_TWINS", "no_list": "RV_NO_LCP_LIST", "no_jump": "RV_NO_TEXT_JUMP", "tab": "RV_DIAG_TABLE", "far_table": "RV_FAR_TABLE", "no_slow": "RV_NO_SLOW_CLASS"}.get(mode)
    if env:
        monkeypatch.setenv(env, "1")
    rng = np.random.default_rng(41)

    def rnd(L):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, L))

    def mutate(s_, every):
        b = bytearray(s_.encode())
        for p in range(every // 2, len(b), every):
            if chr(b[p]) in "ACGT":
                b[p] = ord("ACGT"[("ACGT".index(chr(b[p])) + 1) % 4])
        return b.decode()
    base = rnd(120000)
    elem = rnd(9000)
    rep = base[:20000] + elem + base[20000:50000] + elem + base[50000:70000] + elem[:7000] + base[70000:]      # exact copies of 9 kb and 7 kb
    tandem = base[:30000] + "ACGGTCA" * 1500 + base[30000:60000] + "AC" * 6000 + base[60000:90000]
    many = base[:10000] + ("".join(rnd(40) for _ in range(3)) * 1) .join([elem[:5000]] * 70)                  # 70 copies of a 5 kb element: groups above 64
    near = [g.decode() for g in synth.family(300000, 2, seed=9, snp=0.0002)]
    near_indel = [g.decode() for g in synth.family(200000, 2, seed=10, snp=0.0004, indelfrac=0.3)]
    withn = base[:40000] + "N" * 6000 + base[40000:80000]
    cases = {
        "near": near, "near_indel": near_indel,
        "identical": [base[:60000], base[:60000]],
        "identical_one_snp": [base[:60000], mutate(base[:60000], 45000)],
        "prefix": [base[:60000], base[:52000]],
        "n_run": [withn, mutate(withn, 30000)],                      # a run of N in both, agreement across it
        "repeats": [rep, mutate(rep, 25000)],
        "tandem": [tandem, mutate(tandem, 20000)],
        "many_copies": [many, mutate(many, 50000)],
        "repeats_3": [rep[:60000], mutate(rep[:60000], 9000), mutate(rep[:60000], 7000)],
    }
    far = lst = 0
    for name, seqs in cases.items():
        T, nsep, nodes = assemble(seqs, toupper=False)
        O = oracle(sa64)
        c = O.construct(T, nsep, len(seqs))
        idx = mod(sa64).index()
        for k, s_ in enumerate(seqs):
            idx.addsample("s%d" % k)
            idx.addsequence(s_)
        idx.construct()
        st = idx.sa_stats()
        tag = (name, st)
        assert np.array_equal(idx.array("SA"), c["SA"]), tag
        assert np.array_equal(idx.array("LCP"), c["LCP"]), tag
        assert idx.maxlcp == int(c["LCP"].max()), tag
        if len(seqs) == 2:
            l, a, b = O.getmums(c["tbuf"], c["SA"], c["LCP"], nsep, 20)
            assert idx.getmums(20) == [(int(l[k]), (int(a[k]), int(b[k])), 0) for k in range(len(l))], tag
        got = idx.align_builtin(20, 2)
        ref = O.align_bench(c, nodes, 20, 2)
        gl, goff, gpos = got["anchors"]
        rl, rn, roff, rpos = ref["anchors"]
        assert sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl))) == \
               sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl))), tag
        assert idx.T.encode("latin-1") == ref["T"], tag
        far += st["far_pairs"]; lst += st["lcp_list"]
        if mode == "default":
            if name in ("near", "identical", "identical_one_snp", "prefix"):
                assert st["far_pairs"] > 0 and st["rounds"] <= 2 and st["lcp_list"] == 0, tag      # no doubling round at all
            if name in ("repeats", "tandem", "many_copies"):
                assert st["lcp_list"] > 0, tag
    assert (far > 0) == (mode != "no_far")
    assert (lst > 0) == (mode != "no_list")
