"""Anchors -> variation graph -> GFA1, and back (paths spelled from a GFA file).

The reference turns the recursion's anchors into a graph while it runs (reveal/rem.py:14-382: every chosen
match breaks the nodes it lies in and merges the matched parts) and writes it with utils.write_gfa
(reveal/utils.py:710-839).  For the linear interval model of rem.linear_graphalign the end result has a closed
form, built here after the fact from the anchor list of `index.align_builtin` / `rem.align_genomes`: every anchor
is one node shared by its members, every stretch of a sequence between two anchors is a node of its own, and a
sequence is the path through its pieces.  Record layout as utils.write_gfa: `H VN:Z:1.0`, `S id seq`,
`L from + to + 0M`, `P name id+,id+,... 0M,...` (one path per input sequence).  What the reference's own picker
(schemes.graphmumpicker: chaining, matches present in a subset of the samples) would add is out of scope --
see DESIGN.md, row N3.
"""
import gzip
import sys


def build_graph(text, sequences, anchors):
    """text: the assembled index text (bytes, '$' between sequences); sequences: [(name, (begin, end))] in text
    coordinates; anchors: [(l, (pos, ...))].
    -> (segments: [bytes] with ids 1..n, links: sorted [(from_id, to_id)], paths: [(name, [ids])])"""
    cuts = []                                    # (pos, l, anchor index)
    for k, (l, members) in enumerate(anchors):
        for p in members:
            cuts.append((int(p), int(l), k))
    cuts.sort()
    segments, links, paths = [], set(), []
    anchor_node = {}
    ci = 0
    for name, (b, e) in sorted(sequences, key=lambda x: x[1][0]):
        while ci < len(cuts) and cuts[ci][0] < b:
            ci += 1
        ids, at = [], b
        while ci < len(cuts) and cuts[ci][0] < e:
            p, l, k = cuts[ci]
            if p < at or p + l > e:
                raise ValueError("anchor at %d (length %d) overlaps its neighbour or leaves sequence %s" % (p, l, name))
            if p > at:
                segments.append(text[at:p]); ids.append(len(segments))
            if k not in anchor_node:
                segments.append(text[p:p + l]); anchor_node[k] = len(segments)
            elif segments[anchor_node[k] - 1].upper() != text[p:p + l].upper():
                raise ValueError("members of anchor %d spell different text" % k)
            ids.append(anchor_node[k])
            at = p + l
            ci += 1
        if at < e:
            segments.append(text[at:e]); ids.append(len(segments))
        for u, v in zip(ids, ids[1:]):
            links.add((u, v))
        paths.append((name, ids))
    return segments, sorted(links), paths


def write_gfa(outputfile, segments, links, paths, toupper=True):
    """reveal/utils.py:710-839 record layout"""
    if not outputfile.endswith(".gfa") and not outputfile.endswith(".gfa.gz"):
        outputfile += ".gfa.gz"
    fopen = gzip.open if outputfile.endswith(".gz") else open
    out_of = {}
    for u, v in links:
        out_of.setdefault(u, []).append(v)
    with fopen(outputfile, "wt") as f:
        f.write("H\tVN:Z:1.0\tCL:Z:%s\n" % " ".join(sys.argv))
        for i, seq in enumerate(segments, 1):
            s = seq.decode("latin-1") if isinstance(seq, (bytes, bytearray)) else seq
            f.write("S\t%d\t%s\n" % (i, s.upper() if toupper else s))
            for v in out_of.get(i, ()):
                f.write("L\t%d\t+\t%d\t+\t0M\n" % (i, v))
        for name, ids in paths:
            f.write("P\t%s\t%s\t%s\n" % (name, ",".join("%d+" % i for i in ids), ",".join("0M" for _ in ids)))
    return outputfile


def read_gfa(fn):
    """-> (segments {id: seq}, links [(from, to)], paths [(name, [ids])]) of a GFA1 file with forward-strand paths"""
    fopen = gzip.open if fn.endswith(".gz") else open
    seg, links, paths = {}, [], []
    with fopen(fn, "rt") as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            if c[0] == "S":
                seg[c[1]] = c[2]
            elif c[0] == "L":
                if c[2] != "+" or c[4] != "+":
                    raise ValueError("reverse-strand links are not supported")
                links.append((c[1], c[3]))
            elif c[0] == "P":
                steps = [x for x in c[2].split(",") if x]
                if any(not x.endswith("+") for x in steps):
                    raise ValueError("reverse-strand path steps are not supported")
                paths.append((c[1], [x[:-1] for x in steps]))
    return seg, links, paths


def spell_paths(fn):
    """{path name: sequence} -- the invariant of the reference's test15 (test_reveal.py:150-159): every input
    sequence is spelled by its path; consecutive path steps must be linked"""
    seg, links, paths = read_gfa(fn)
    ls = set(links)
    out = {}
    for name, ids in paths:
        for u, v in zip(ids, ids[1:]):
            if (u, v) not in ls:
                raise ValueError("path %s steps from %s to %s without a link" % (name, u, v))
        out[name] = "".join(seg[i] for i in ids)
    return out
