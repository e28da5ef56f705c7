"""`reveal rem` for Python 3 around the `index` type (reveal/rem.py, reveal/utils.py, reveal/schemes.py).

Two drivers share the index protocol of rem.py:511-611 (`addsample` per file, `addsequence` per contig / per graph
node, `construct`, `align` with two callbacks):

  * `graph_rem(inputs, output)` -- the reference's own pair of callbacks: `GraphAligner.graphalign` (rem.py:318-382:
    break the nodes under the match, merge them, segment the graph into leading / trailing / parallel intervals) and
    `schemes.GraphPicker.graphmumpicker` (schemes.py:197-361: filter, trim, map to path offsets, chain -- the DP in C++
    behind the ABI -- split on the largest match, seed the children).  Inputs are FASTA files and / or GFA graphs written
    by earlier runs (one sample per file, one '$'-terminated sequence per S-line: utils.py:377-677), the output is GFA1
    (utils.py:710-839) -- what `reveal align --order=sequential` chains level by level (reveal_amd/align.py).
  * `rem(inputs, output)` -- the deterministic benchmark callbacks of SURVEY.md 8(d) (longest full match, linear interval
    model), which the library also has built in (`index.align_builtin`): what bench.py times and the parity tests trace.
"""
import gzip
import os
import sys


_ODD_LINE_END = None


def _fasta_records_fast(data, toupper, keepdash):
    """the records of a FASTA file held as bytes, by whole-buffer operations (a 5 Mbp genome line by line through str methods: 60 ms; this way: 6) -- or None when a line ends in
    white space other than its newline (the line-by-line reader strips that, too: it takes the file then)"""
    global _ODD_LINE_END
    import re
    if _ODD_LINE_END is None:
        _ODD_LINE_END = re.compile(rb"[ \t\x0b\x0c\r](?:\n|$)")
    if b"\r" in data or _ODD_LINE_END.search(data) or not data.startswith(b">"):      # (a carriage return anywhere: text mode makes a line break of it)
        return None
    out = []
    drop = b"\n" if keepdash else b"\n-"
    recs = data.split(b"\n>")
    for k, rec in enumerate(recs):
        head, sep, body = rec.partition(b"\n")
        # (the line-by-line reader yields a record that holds at least one line behind its header, however empty: a record in front of another one lost its last
        #  newline to the split, the last one's trailing newline ends its header)
        has_lines = bool(body) or (k + 1 < len(recs) and bool(sep))
        if head.startswith(b">"):
            head = head[1:]
        seq = body.translate(None, drop)
        if b">" in seq:      # (a '>' inside a line of sequence: not a header for the line-by-line reader either -- leave the file to it)
            return None
        if toupper:
            seq = seq.upper()
        if has_lines:
            out.append((head.replace(b">", b"").replace(b"\t", b"").decode("utf-8"), seq.decode("utf-8")))
    return out


def fasta_reader(fn, toupper=True, keepdash=False):
    """reveal/utils.py:79-160 with its defaults (truncN=False, cutN=0)."""
    fopen = gzip.open if fn.endswith(".gz") else open
    try:
        with fopen(fn, "rb") as fb:
            recs = _fasta_records_fast(fb.read(), toupper, keepdash)
    except (UnicodeDecodeError, MemoryError):
        recs = None
    if recs is not None:
        for name, seq in recs:
            yield name, seq
        return
    name, seq = None, []
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


class GraphAligner:
    """graphalign bound to an alignment graph (reveal_amd/alngraph.py); rem.py:318-382"""

    def __init__(self, graph):
        self.G = graph
        self.calls = 0

    def graphalign(self, index, mum):
        G = self.G
        self.calls += 1
        l, n, spd = mum
        nodes = index.nodes                        # the set handed out by an earlier call: edited in place, as in the reference
        mns, matching = [], set()
        for _, pos in spd:
            matching.add((pos, pos + l))
            old = G.node_at(pos)
            mn, other = G.breaknode(old, pos, l)
            mns.append(mn)
            nodes.remove(old)
            nodes.update(other)
        mn = G.mergenodes(mns)
        msamples = set(G.offsets[mn])
        leading, trailing, rest = G.segmentgraph(mn, nodes)
        newleft = newright = mn
        if any(not set(G.offsets[iv]) <= msamples for iv in leading):      # no clean dissection of all paths on the left
            newright = index.rightnode
        if any(not set(G.offsets[iv]) <= msamples for iv in trailing):
            newleft = index.leftnode
        # (matching as a list in ascending order: the index walks it when it shortens the leading child's suffixes at the cuts, reveal.c:673-674, and the
        #  order of equal truncated suffixes -- with it the order a later match's members are emitted in, which trim_overlap's coordinate-wise cuts
        #  look at -- follows that walk.  The reference hands over a set, i.e. its hash order: any fixed order is as good; this one is the library's
        #  own (rv_set_picker), so the two ways of running the picker give the same graph.)
        return leading, trailing, sorted(matching), rest, mn, newleft, newright


class _ReplaySub:
    """what graphalign reads and edits of a sub-index (rem.py:318-382): its set of intervals and its left / right graph nodes"""
    __slots__ = ("nodes", "leftnode", "rightnode", "depth")

    def __init__(self, nodes, leftnode, rightnode, depth):
        self.nodes, self.leftnode, self.rightnode, self.depth = nodes, leftnode, rightnode, depth


def replay_anchors(G, aligner, root_nodes, anchors):
    """The graph of a run whose picker ran inside the library (index.set_picker + align_builtin): the recursion's bookkeeping of
    index.align (level by level; children: leading, trailing, rest -- reveal.c:1136-1207) with graphalign applied to every anchor in the
    order the library chose them.  anchors: [(l, n, ((sample, pos), ...))] in emission order.  -> number of graphalign calls"""
    frontier = [_ReplaySub(set(root_nodes), None, None, 0)]
    ai, na = 0, len(anchors)
    while frontier and ai < na:
        nxt = []
        for sub in frontier:
            if ai >= na:
                break
            mum = anchors[ai]
            p0 = mum[2][0][1]
            if not any(b <= p0 < e for b, e in sub.nodes):
                continue
            ai += 1
            r = aligner.graphalign(sub, mum)
            if r is None:
                continue
            leading, trailing, matching, rest, merged, newleft, newright = r
            if leading:
                nxt.append(_ReplaySub(leading, sub.leftnode, newright, sub.depth + 1))
            if trailing:
                nxt.append(_ReplaySub(trailing, newleft, sub.rightnode, sub.depth + 1))
            if rest:
                nxt.append(_ReplaySub(rest, sub.leftnode, sub.rightnode, sub.depth + 1))
        frontier = nxt
    if ai != na:
        raise RuntimeError("replay_anchors: %d of %d anchors found no sub-index" % (na - ai, na))
    return na


def replay_anchors_fast(G, aligner, root_nodes, anchors):
    """replay_anchors without the recursion's bookkeeping: what graphalign does to the GRAPH for an anchor -- break the nodes that hold its members,
    merge the pieces (rem.py:331-345) -- depends on the graph alone, and replay_anchors applies the anchors strictly in the order given; the
    sub-index' interval set, its left / right nodes and segmentgraph's walks only serve the index, which has already finished.  Same graph, node
    for node and edge for edge in the same insertion order (the GFA writer's order)."""
    node_at, breaknode, mergenodes = G.node_at, G.breaknode, G.mergenodes
    for l, n, spd in anchors:
        mergenodes([breaknode(node_at(pos), pos, l)[0] for _, pos in spd])
    aligner.calls += len(anchors)
    return len(anchors)


class _Stages:
    """REVEAL_AMD_TIMES=1: wall-clock seconds of a job's stages on stderr (reading, index, recursion, graph, file)"""
    on = os.environ.get("REVEAL_AMD_TIMES", "0") not in ("0", "")

    def __init__(self):
        import time
        self.clock, self.t, self.rows = time.perf_counter, time.perf_counter(), []

    def mark(self, what):
        if self.on:
            now = self.clock()
            self.rows.append((what, now - self.t))
            self.t = now

    def report(self, extra=""):
        if self.on and self.rows:
            sys.stderr.write("stages: " + ", ".join("%s %.3f s" % r for r in self.rows) + (" | " + extra if extra else "") + "\n")
            self.rows = []


_stages = _Stages()


def graph_align_genomes(inputfiles, sa64=False, minlength=20, minn=2, contigs=True, toupper=True, args=None, preselect=True, indexmod=None, native=None, materialize=True):
    """rem.py:511-611 align_genomes: index + graph from FASTA / GFA inputs, construct, align with the graph callbacks.
    preselect: let the library hand the picker only what it keeps anyway (index.preselect(maxmums): the matches spanning
    every sample of the sub-index, capped at --maxmums) -- not valid with --trim, which looks at the others
    indexmod: the module that provides `index` (default reveal_amd.reveallib / reveallib64; tests pass the reference's own module)
    native: both callbacks inside the library -- FASTA inputs with one sequence per sample: the picker (rv_set_picker), the graph from the run's anchors
    afterwards (alngraph.NativeGraph); graphs / several sequences per sample: readers, picker and graphalign on the graph behind the ABI (alngraph.LoopGraph,
    rv_set_graph_picker).  None = wherever the inputs allow it, False = the Python callbacks.  materialize=False leaves the graph behind the ABI --
    `graph.native` holds it, the returned graph is the reader's (only its path tables when the inputs were read there) -- for a caller that only wants the file
    -> (graph, index, picker, aligner)"""
    from . import alngraph, schemes
    if indexmod is None:
        from . import reveallib, reveallib64
        indexmod = reveallib64 if sa64 else reveallib
    idx = indexmod.index()
    G = alngraph.AlnGraph()
    args = args or schemes.PickerArgs()
    env_on = os.environ.get("REVEAL_AMD_NATIVE", "1") not in ("0", "false", "no", "off")
    gfa_in = any(f.endswith(".gfa") or f.endswith(".gfa.gz") for f in inputfiles)
    loop_ok = native is not False and hasattr(idx, "set_graph_picker") and args.maxsize is None and args.maxdepth is None and (native or env_on)
    loop = None
    if gfa_in and loop_ok:
        # graphs among the inputs: read behind the ABI as well (csrc/rv_gfaread.hip; the Python reader took 9.3 of the 14.8 s of a job of five 5 x 1 Mbp graphs)
        try:
            loop = alngraph.LoopGraph.read(inputfiles, idx, G, contigs=contigs, toupper=toupper, sa64=sa64)
        except alngraph.ReverseStrand:
            idx, G, loop = indexmod.index(), alngraph.AlnGraph(), None      # (half-filled: start over)
            loop_ok = False
    if loop is None:
        if getattr(idx, "_h", None) is not None and not any(f.endswith(".gz") for f in inputfiles):
            # one allocation of the (page-locked) host text instead of one per sequence: a file is never shorter than the text it adds
            idx._dll.rv_reserve_text(idx._h, int(idx._dll.rv_n(idx._h)) + sum(os.path.getsize(f) for f in inputfiles) + 64)
        for f in inputfiles:
            if f.endswith(".gfa") or f.endswith(".gfa.gz"):
                idx.addsample(os.path.basename(f))
                alngraph.read_gfa(f, idx, G)
            else:
                alngraph.read_fasta(f, idx, G, contigs=contigs, toupper=toupper)
    if len(idx.samples) <= 1:
        raise ValueError("Specify at least 2 targets to construct alignment. In case of multi-fasta, consider the --nocontigs flag.")
    _stages.mark("read inputs")
    picker, aligner = schemes.GraphPicker(G, args), GraphAligner(G)
    # native: the picker inside the library (rv_set_picker / rv_pick_chain: no Python call per sub-index), the graph from the anchors afterwards.
    # Its case: FASTA inputs with one sequence per sample, no --maxbubblesize / maxdepth, reveal_amd's own index.  None = whenever that holds.
    can_native = (hasattr(idx, "set_picker") and not gfa_in and len(idx.nodes) == len(idx.samples) and args.maxsize is None and args.maxdepth is None)
    # The other inputs -- graphs (GFA files of earlier alignments: read behind the ABI above), samples of several sequences (the readers' graph moved there:
    # alngraph.LoopGraph): picker AND graphalign inside the library (rv_set_graph_picker); links on the reverse strand keep the Python callbacks.
    if loop is None and not can_native and loop_ok:
        try:
            loop = alngraph.LoopGraph(G, sa64=sa64)
        except ValueError:
            loop = None
    if native and not can_native and loop is None:
        raise ValueError("native=True: no links on the reverse strand, no maxsize / maxdepth, reveal_amd's index")
    if native is None:      # (REVEAL_AMD_NATIVE=0 in the environment, or --no-native on the command line: the Python callbacks)
        native = can_native and env_on
    _stages.mark("graph behind the ABI" if loop is not None else "setup")
    root_nodes = sorted(tuple(x) for x in idx.nodes) if loop is None else None
    idx.construct()
    _stages.mark("construct")
    if loop is not None:
        idx.set_graph_picker(loop, args)
        try:
            res = idx.align_builtin(minlength, minn)
        finally:
            idx.set_graph_picker(None)
        _stages.mark("recursion")
        loop.finish()
        _stages.mark("renumber")
        picker.calls = idx.picker_info()["calls"]
        _stages.report("picker %(calls)d calls (%(seeded)d seeded), pick %(picker_s).3f s, lists %(lists_s).3f s, graphalign %(graphalign_s).3f s" % idx.picker_info())
        aligner.calls += len(res["anchors"][0])
        G.native = loop
        if materialize:
            loop.load_into(G)
            loop.close()
            G.native = None
        return G, idx, picker, aligner
    if native:
        # graphalign's surgery for every anchor, in the order the library chooses them, on a host thread that follows the run level by level (rv_set_replay_graph;
        # replay_anchors / replay_anchors_fast are the same in Python, after the run: 25 s instead of 1.1 for five genomes of 5 Mbp)
        idx.set_picker(args)
        ng = alngraph.NativeGraph(G, root_nodes, sa64=sa64)
        idx.set_replay_graph(ng)
        try:
            l, off, pos = idx.align_builtin(minlength, minn)["anchors"]
        finally:
            idx.set_replay_graph(None)
            idx.set_picker(None)
        _stages.mark("recursion + graph")
        picker.calls = idx.picker_info()["calls"]
        idx._nodes = set(root_nodes)
        G.native = ng
        aligner.calls += len(l)
        if materialize:
            G.native.load_into(G)
            G.native.close()
            G.native = None
        return G, idx, picker, aligner
    if preselect and not args.trim and args.maxmums and hasattr(idx, "preselect"):
        idx.preselect(args.maxmums)
    idx.align(picker.graphmumpicker, aligner.graphalign, threads=0, wpen=args.wpen, wscore=args.wscore, minl=minlength, minn=minn)
    return G, idx, picker, aligner


def align(aobjs, ref=None, minlength=20, minn=2, seedsize=None, threads=0, targetsample=None, maxsamples=None,
          maxmums=10000, wpen=1, wscore=1, sa64=False, pcutoff=1e-8, gcmodel="sumofpairs", maxsize=None, trim=True, indexmod=None):
    """reveal/rem.py:616-712 `align(aobjs, ...)` -> (G, idx): the alignment of sequences given as (name, sequence) tuples, as the
    reference's refine step calls it for the sequences of a bubble (reveal/refine.py:220-229) and its test01 for two 17-mers
    (reveal/tests/test_reveal.py:36-41).  One sample per tuple, upper-cased, empty sequences skipped; every sequence node hangs
    between ONE start and ONE end sentinel (the reference's startnode / endnode), which are removed again before the graph is
    returned, after prune_nodes (rem.py:706-710).  Same keyword arguments and defaults as the reference; `ref`, `threads`,
    `targetsample`, `maxsamples` are accepted and, like there, not used by this path.
    G is an alngraph.AlnGraph (number_of_nodes() / number_of_edges() as in networkx); idx the index, its T lower-cased where aligned."""
    from . import alngraph, schemes
    if indexmod is None:
        from . import reveallib, reveallib64
        indexmod = reveallib64 if sa64 else reveallib
    idx = indexmod.index()
    G = alngraph.AlnGraph()
    import uuid
    startnode, endnode = uuid.uuid4().hex, uuid.uuid4().hex
    G.add_node(startnode, offsets={})
    G.add_node(endnode, offsets={})
    G.startnodes.append(startnode)
    G.endnodes.append(endnode)
    for aobj in aobjs:
        if not isinstance(aobj, tuple):
            continue                                  # (the reference only handles tuples here, rem.py:647-648)
        name, seq = aobj
        idx.addsample(name)
        intv = tuple(idx.addsequence(seq.upper()))
        if intv[1] - intv[0] > 0:
            sid = alngraph._new_path(G, name, len(seq))
            G.add_node(intv, offsets={sid: 0}, aligned=0)
            G.offsets[startnode][sid] = 0
            G.offsets[endnode][sid] = len(seq)
            G.add_edge(startnode, intv, {sid})
            G.add_edge(intv, endnode, {sid})
    if len(G.paths) < 2:
        raise ValueError("Specify at least 2 targets to construct alignment.")
    args = schemes.PickerArgs(wscore=wscore, wpen=wpen, maxmums=maxmums, seedsize=seedsize if seedsize is not None else 10000, gcmodel=gcmodel,
                              trim=trim, maxsize=maxsize, pcutoff=pcutoff)
    picker, aligner = schemes.GraphPicker(G, args), GraphAligner(G)
    idx.construct()
    idx.align(picker.graphmumpicker, aligner.graphalign, threads=threads, wpen=wpen, wscore=wscore, minl=minlength, minn=minn)
    G.prune_nodes(idx.T)
    G.remove_node(startnode)
    G.remove_node(endnode)
    G.startnodes.remove(startnode)
    G.endnodes.remove(endnode)
    return G, idx


def graph_rem(inputfiles, output=None, sa64=False, minlength=20, minn=2, contigs=True, toupper=True, args=None, preselect=True, indexmod=None, native=None, materialize=True):
    """`reveal rem inputs -o output` (rem.py:449-509 align_cmd): align, merge equal siblings when more than two paths took part,
    write GFA1.  -> (graph, index, file name or None)
    materialize=False: when the graph was built behind the ABI (native picker) it is pruned and written there and not turned into Python
    objects (most of what the graph then takes): in its place comes {"seq_nodes": .., "edges": .., "paths": [names]}"""
    from . import alngraph
    G, idx, picker, aligner = graph_align_genomes(inputfiles, sa64=sa64, minlength=minlength, minn=minn, contigs=contigs, toupper=toupper,
                                                  args=args, preselect=preselect, indexmod=indexmod, native=native, materialize=False)
    T = idx.T
    ng = getattr(G, "native", None)
    if ng is not None:
        Tb = T.encode("latin-1")
        _stages.mark("text")
        if len(G.paths) > 2:
            ng.prune(Tb)
        _stages.mark("prune")
        fn = ng.write_gfa(Tb, output, cmdline="reveal_amd.rem " + " ".join(inputfiles)) if output else None
        _stages.mark("gfa")
        _stages.report()
        summary = dict(zip(("seq_nodes", "edges"), ng.counts()), paths=list(G.paths))
        if materialize:
            ng.load_into(G)
        ng.close()
        G.native = None
        return (G if materialize else summary), idx, fn
    if len(G.paths) > 2:
        G.prune_nodes(T)
    fn = None
    if output:
        fn = alngraph.write_gfa(G, T, output, cmdline="reveal_amd.rem " + " ".join(inputfiles))
    return G, idx, fn


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
    ap.add_argument("--bench-callbacks", action="store_true", help="the deterministic benchmark callbacks (FASTA inputs only) instead of the reference's graph callbacks")
    ap.add_argument("--wp", dest="wpen", type=int, default=1)
    ap.add_argument("--ws", dest="wscore", type=int, default=1)
    ap.add_argument("--seedsize", type=int, default=10000)
    ap.add_argument("--maxmums", type=int, default=1000)
    ap.add_argument("--gcmodel", choices=["sumofpairs", "star-avg", "star-med"], default="sumofpairs")
    ap.add_argument("--notrim", dest="trim", action="store_false")
    ap.add_argument("--maxbubblesize", dest="maxsize", type=int, default=None)
    ap.add_argument("-p", dest="pcutoff", type=float, default=1e-8)
    ap.add_argument("--no-native", dest="native", action="store_const", const=False, default=None,
                    help="run the reference's graphmumpicker / graphalign as Python callbacks per sub-index instead of the C++ picker and graph behind the ABI "
                         "(the default wherever the inputs allow it; REVEAL_AMD_NATIVE=0 does the same)")
    a = ap.parse_args(argv)
    if a.bench_callbacks:
        idx, (segments, links, paths), fn = rem(a.inputfiles, a.output, sa64=a.sa64, minlength=a.minlength, minn=a.minn, contigs=a.contigs)
        print("%s: %d segments, %d links, %d paths" % (fn, len(segments), len(links), len(paths)))
        return
    from . import schemes
    pa = schemes.PickerArgs(wscore=a.wscore, wpen=a.wpen, maxmums=a.maxmums, seedsize=a.seedsize, gcmodel=a.gcmodel, trim=a.trim, maxsize=a.maxsize, pcutoff=a.pcutoff)
    G, idx, fn = graph_rem(a.inputfiles, a.output, sa64=a.sa64, minlength=a.minlength, minn=a.minn, contigs=a.contigs, args=pa, native=a.native, materialize=False)
    if isinstance(G, dict):      # (built, pruned and written behind the ABI)
        print("%s: %d nodes, %d paths" % (fn, G["seq_nodes"], len(G["paths"])))
    else:
        print("%s: %d nodes, %d paths" % (fn, len(G.seq_nodes()), len(G.paths)))


if __name__ == "__main__":
    main()
