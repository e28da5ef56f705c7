"""Python-3 harness around the `index` type: the call protocol of the
reference's `reveal rem` driver (reveal/rem.py:511-611 align_genomes,
reveal/utils.py:304-375 read_fasta) plus the deterministic benchmark callbacks
of SURVEY.md 8(d).  The reference's graph layer (networkx graph, interval tree,
chaining picker) is out of scope; this module feeds the index the way the
reference does, supplies callbacks with the reference's signatures, and -- for the
linear interval model of those callbacks -- writes the resulting graph as GFA1
(reveal_amd/gfa.py; `python -m reveal_amd.rem a.fa b.fa -o out.gfa`).
"""
import gzip
import os


def fasta_reader(fn, toupper=True, keepdash=False):
    """reveal/utils.py:79-160 with its defaults (truncN=False, cutN=0)."""
    name, seq = None, []
    fopen = gzip.open if fn.endswith(".gz") else open
    with fopen(fn, "rt") as ff:
        for line in ff:
            line = line.rstrip()
            if line.startswith(">"):
                if seq:
                    yield name, "".join(seq)
                name, seq = line.replace(">", "").replace("\t", ""), []
            else:
                if toupper:
                    line = line.upper()
                if not keepdash:
                    line = line.replace("-", "")
                seq.append(line)
    if seq:
        yield name, "".join(seq)


def read_fasta(fasta, index, contigs=True, toupper=True):
    """reveal/utils.py:304-375 minus the graph bookkeeping: one addsample per
    file and one addsequence per contig (or, with contigs=False, one sample per
    sequence).  Returns [(name, (begin, end))]."""
    out = []
    if contigs:
        index.addsample(os.path.basename(fasta))
        for name, seq in fasta_reader(fasta, toupper=toupper):
            out.append((name, index.addsequence(seq)))
    else:
        for name, seq in fasta_reader(fasta, toupper=toupper):
            index.addsample(name)
            out.append((name, index.addsequence(seq)))
    return out


def add_sequences(index, seqs, names=None):
    """rem.align's protocol (reveal/rem.py:647-671): one sample per sequence."""
    out = []
    for k, s in enumerate(seqs):
        index.addsample(names[k] if names else "s%d" % k)
        out.append(index.addsequence(s if isinstance(s, (bytes, bytearray)) else s.upper()))
    return out


# ---- benchmark callbacks (same contracts as schemes.graphmumpicker / rem.graphalign)

def bench_mumpicker(mums, idx, precomputed=False, minlength=0):
    """mumpicker(mums, idx, precomputed=, minlength=) -> () | (mum, skipleft, skipright)
    (reveal.c:839-895).  Keeps matches present in every sample of the
    sub-index (schemes.py:227), takes the longest, ties -> smallest minimum
    coordinate; no seeds."""
    best = None
    ns = idx.nsamples
    for m in mums:
        if m[1] != ns:
            continue
        mn = min(p for _, p in m[2])
        if best is None or m[0] > best[0][0] or (m[0] == best[0][0] and mn < best[1]):
            best = (m, mn)
    if best is None:
        return ()
    return (best[0], [], [])


def seeding_mumpicker(mums, idx, precomputed=False, minlength=0):
    """bench_mumpicker that also seeds the children the way schemes.graphmumpicker does with its chain
    (schemes.py:291-361: `return splitmum, skipleft, skipright`): of the full matches it was given, those lying in front of the
    pick in every sample go to the leading child as `skipmums`, those behind it to the trailing child; a child that
    received a non-empty list is not scanned (reveal.c:802, 830-837) and is called with precomputed=True and that list."""
    r = bench_mumpicker(mums, idx, precomputed=precomputed, minlength=minlength)
    if not r:
        return ()
    pick = r[0]
    l = pick[0]
    at = {so: p for so, p in pick[2]}
    ns = idx.nsamples
    left, right = [], []
    for m in mums:
        if m is pick or m[1] != ns or m[0] < max(minlength, 1):
            continue
        if all(so in at for so, _ in m[2]):
            if all(p + m[0] <= at[so] for so, p in m[2]):
                left.append(m)
            elif all(p >= at[so] + l for so, p in m[2]):
                right.append(m)
    return (pick, left, right)


def linear_graphalign(idx, mum):
    """graphalign(idx, mum) -> (leading, trailing, matching, rest, merged, newleft, newright)
    (reveal.c:937-999) for the linear interval model: every member lies in one
    interval of the sub-index; left remainders lead, right remainders trail."""
    l, n, spd = mum
    nodes = sorted(idx.nodes)
    lead, trail, match, touched = [], [], [], set()
    for _, sp in spd:
        hit = None
        for q, (b, e) in enumerate(nodes):
            if b <= sp < e:
                hit = q
                break
        if hit is None or sp + l > nodes[hit][1]:
            return None
        b, e = nodes[hit]
        touched.add(hit)
        if sp > b:
            lead.append((b, sp))
        if sp + l < e:
            trail.append((sp + l, e))
        match.append((sp, sp + l))
    rest = [iv for q, iv in enumerate(nodes) if q not in touched]
    merged = tuple(sorted(match)[0])
    return sorted(lead), sorted(trail), sorted(match), sorted(rest), merged, merged, merged


def align_genomes(inputfiles, sa64=False, minlength=20, minn=2, contigs=True, toupper=True, sa="", lcp="", cache=0,
                  mumpicker=bench_mumpicker, graphalign=linear_graphalign):
    """reveal/rem.py:511-611: build the index from FASTA files, construct, align.
    Returns (idx, anchors) where anchors is the list of matches handed to
    graphalign, in callback order."""
    from . import reveallib, reveallib64
    mod = reveallib64 if sa64 else reveallib
    idx = mod.index(sa=sa, lcp=lcp, cache=cache)
    for f in inputfiles:
        read_fasta(f, idx, contigs=contigs, toupper=toupper)
    if len(idx.samples) <= 1:
        raise ValueError("Specify at least 2 targets to construct alignment.")
    idx.construct()
    anchors = []

    def galign(i, mum):
        r = graphalign(i, mum)
        if r is not None:
            anchors.append(mum)
        return r
    idx.align(mumpicker, galign, threads=0, wpen=1, wscore=1, minl=minlength, minn=minn)
    return idx, anchors


def rem(inputfiles, output=None, sa64=False, minlength=20, minn=2, contigs=True, toupper=True, builtin=True):
    """`reveal rem` for FASTA inputs with the benchmark callbacks: index, align, graph, GFA.
    builtin=True runs the recursion without Python in the loop (index.align_builtin), False through align() and the
    two Python callbacks -- same anchors.  -> (index, (segments, links, paths), path of the GFA file or None)"""
    from . import gfa, reveallib, reveallib64
    mod = reveallib64 if sa64 else reveallib
    idx = mod.index()
    seqs = []
    for f in inputfiles:
        seqs += read_fasta(f, idx, contigs=contigs, toupper=toupper)
    if len(idx.samples) <= 1:
        raise ValueError("Specify at least 2 targets to construct alignment.")
    text = idx.T.encode("latin-1")                       # before align() lower-cases the matched parts
    idx.construct()
    if builtin:
        l, off, pos = idx.align_builtin(minlength, minn)["anchors"]
        anchors = [(int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l))]
    else:
        picked = []

        def galign(i, mum):
            r = linear_graphalign(i, mum)
            if r is not None:
                picked.append(mum)
            return r
        idx.align(bench_mumpicker, galign, threads=0, wpen=1, wscore=1, minl=minlength, minn=minn)
        anchors = [(m[0], tuple(p for _, p in m[2])) for m in picked]
    graph = gfa.build_graph(text, seqs, anchors)
    fn = gfa.write_gfa(output, *graph, toupper=toupper) if output else None
    return idx, graph, fn


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m reveal_amd.rem", description="recursive exact matching of FASTA files on an MI355X -> GFA1")
    ap.add_argument("inputfiles", nargs="+")
    ap.add_argument("-o", "--output", default="reveal_amd.gfa")
    ap.add_argument("-m", dest="minlength", type=int, default=20)
    ap.add_argument("-n", dest="minn", type=int, default=2)
    ap.add_argument("--64", dest="sa64", action="store_true")
    ap.add_argument("--nocontigs", dest="contigs", action="store_false")
    a = ap.parse_args(argv)
    idx, (segments, links, paths), fn = rem(a.inputfiles, a.output, sa64=a.sa64, minlength=a.minlength, minn=a.minn, contigs=a.contigs)
    print("%s: %d segments, %d links, %d paths" % (fn, len(segments), len(links), len(paths)))


if __name__ == "__main__":
    main()
