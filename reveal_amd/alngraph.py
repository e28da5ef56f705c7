"""The alignment graph of `reveal rem` for Python 3 (reveal/rem.py:14-382, reveal/utils.py:304-677, 710-839).

The reference keeps a networkx MultiDiGraph whose sequence nodes are intervals of the index text, plus an interval tree for
"which node holds text position p".  Neither library is needed for what the recursion's callbacks do, so this module has
its own small structure:

  * a sequence node is the tuple (begin, end) of its text interval -- the very objects the index hands to and takes from the
    callbacks (`idx.nodes`, the interval lists returned by graphalign); sentinels (start / end of a path or component) are str
  * offsets[node] = {path id: offset of the node on that path}, aligned[node] = 0 / 1 (sentinels have no entry)
  * succ[u][(v, ofrom, oto)] = set of path ids, pred[v][(u, ofrom, oto)] = the same set object
  * nodes never overlap in the text and are only ever split, so position lookup is a bisect over the sorted node begins

Functions follow the reference's semantics (cited at each one); the code is not a translation.
"""
import bisect
import gzip
import os
import sys
import uuid

class _Begins:
    """the node begins, sorted, for "which node holds text position p": blocks of a few hundred values and the blocks' largest values, both searched
    with bisect (a begin is only ever added: nodes are split, never joined across a begin).  What this file needs of a sorted list and nothing else --
    the general-purpose one it used before spent 11 us per lookup in Python-level calls, this one 2; 15 % of a merge of graphs"""
    LOAD = 512

    def __init__(self, values=()):
        values = sorted(values)
        self._lists = [values[i:i + self.LOAD] for i in range(0, len(values), self.LOAD)]
        self._maxes = [b[-1] for b in self._lists]

    def add(self, x):
        lists, maxes = self._lists, self._maxes
        if not maxes:
            lists.append([x]); maxes.append(x)
            return
        k = bisect.bisect_left(maxes, x)
        if k == len(maxes):
            k -= 1
            lists[k].append(x); maxes[k] = x
        else:
            bisect.insort(lists[k], x)
        if len(lists[k]) > 2 * self.LOAD:
            blk = lists[k]
            half = blk[self.LOAD:]
            del blk[self.LOAD:]
            maxes[k] = blk[-1]
            lists.insert(k + 1, half); maxes.insert(k + 1, half[-1])

    def pred(self, x):
        """the largest value <= x, None when there is none"""
        maxes = self._maxes
        k = bisect.bisect_left(maxes, x)              # the first block whose largest value is >= x
        if k < len(maxes):
            blk = self._lists[k]
            j = bisect.bisect_right(blk, x)
            if j:
                return blk[j - 1]
        return self._lists[k - 1][-1] if k else None

    def __len__(self):
        return sum(len(b) for b in self._lists)

    def __iter__(self):
        for b in self._lists:
            yield from b


class AlnGraph:
    def __init__(self):
        self.offsets = {}          # node -> {sid: offset}; insertion order = node order of the GFA writer
        self.aligned = {}          # sequence nodes only
        self.seq = {}              # sentinels / nodes that carry their own sequence ("" for merged start / end nodes)
        self.succ, self.pred = {}, {}
        self.paths, self.path2id, self.id2path, self.id2end = [], {}, {}, {}
        self.startnodes, self.endnodes = [], []
        self.literal_segments = False      # segmentgraph as the reference spells it (set by check_segment_shortcut)
        self._begins = _Begins()
        self._end_of = {}
        self.native = None          # NativeGraph: the run's graph while it is kept behind the ABI (rem.graph_align_genomes(materialize=False))

    # ---- nodes and edges -------------------------------------------------------------------
    def add_node(self, node, offsets=None, aligned=None, seq=None):
        self.offsets[node] = offsets if offsets is not None else {}
        self.succ.setdefault(node, {})
        self.pred.setdefault(node, {})
        if aligned is not None:
            self.aligned[node] = aligned
        if seq is not None:
            self.seq[node] = seq
        if isinstance(node, tuple):
            if node[0] not in self._end_of:
                self._begins.add(node[0])
            self._end_of[node[0]] = node[1]

    def remove_node(self, node):
        for (v, a, b) in list(self.succ[node]):
            del self.pred[v][(node, a, b)]
        for (u, a, b) in list(self.pred[node]):
            del self.succ[u][(node, a, b)]
        del self.succ[node], self.pred[node], self.offsets[node]
        self.aligned.pop(node, None)
        self.seq.pop(node, None)
        if isinstance(node, tuple) and self._end_of.get(node[0]) == node[1]:
            self._end_of[node[0]] = node[0]          # an empty interval: lookups fall through

    def add_edge(self, u, v, paths, ofrom="+", oto="+"):
        """one edge per (u, v, ofrom, oto); adding it again merges the path sets"""
        key = (v, ofrom, oto)
        if key in self.succ[u]:
            self.succ[u][key] |= paths
        else:
            self.succ[u][key] = paths
            self.pred[v][(u, ofrom, oto)] = paths

    def has_node(self, node):
        return node in self.offsets

    def number_of_nodes(self):
        return len(self.offsets)

    def number_of_edges(self):
        return sum(len(v) for v in self.succ.values())

    def seq_nodes(self):
        return [n for n in self.offsets if isinstance(n, tuple)]

    def node_at(self, pos):
        """the sequence node whose text interval holds pos (the reference's `t[pos]`, rem.py:334)"""
        b = self._begins.pred(pos)
        if b is not None:
            e = self._end_of[b]
            if pos < e and (b, e) in self.offsets:
                return (b, e)
        raise KeyError("no node holds text position %d" % pos)

    def _real(self, paths):
        return any(not self.id2path[p].startswith("*") for p in paths)

    # ---- callbacks' graph surgery ----------------------------------------------------------------
    def breaknode(self, node, pos, l):
        """rem.py:14-131: cut [pos, pos+l) out of `node`; -> (match node, set of the prefix / suffix nodes created)"""
        mn = (pos, pos + l)
        other = set()
        if mn == node:
            return node, other
        att = self.offsets[node]
        in_edges = [(u, a, b, p) for (u, a, b), p in self.pred[node].items()]
        out_edges = [(v, a, b, p) for (v, a, b), p in self.succ[node].items()]
        moffsets = {s: o + (pos - node[0]) for s, o in att.items()}
        soffsets = {s: o + (pos + l - node[0]) for s, o in att.items()}
        negpaths, pospaths = set(), set()
        negstrand = False
        if not in_edges and not out_edges:
            pospaths = set(att)
        else:
            for u, a, b, p in in_edges:               # the strand a path enters the node on
                if b == "-":
                    negstrand = True
                    negpaths |= p
                else:
                    pospaths |= p
            for v, a, b, p in out_edges:
                if a == "-":
                    negstrand = True
                    negpaths |= p
                else:
                    pospaths |= p
        if pospaths & negpaths:
            raise ValueError("paths traverse node %s on both strands: cannot break it" % (node,))
        self.add_node(mn, offsets=moffsets, aligned=0)
        pn = sn = mn
        if node[0] != pos:
            pn = (node[0], pos)
            self.add_node(pn, offsets=dict(att), aligned=0)
            self.add_edge(pn, mn, set(pospaths), "+", "+")
            if negstrand:
                self.add_edge(mn, pn, set(negpaths), "-", "-")
            other.add(pn)
        if node[1] != pos + l:
            sn = (pos + l, node[1])
            self.add_node(sn, offsets=soffsets, aligned=0)
            self.add_edge(mn, sn, set(pospaths), "+", "+")
            if negstrand:
                self.add_edge(sn, mn, set(negpaths), "-", "-")
            other.add(sn)
        self.remove_node(node)
        if mn[0] == node[0]:                          # (remove_node blanked the begin the match / prefix node shares with the old node)
            self._end_of[mn[0]] = mn[1]
        if pn is not mn:
            self._end_of[pn[0]] = pn[1]
        for u, a, b, p in in_edges:
            self.add_edge(u, pn if b == "+" else sn, p, a, b)
        for v, a, b, p in out_edges:
            self.add_edge(sn if a == "+" else pn, v, p, a, b)
        return mn, other

    def mergenodes(self, mns):
        """rem.py:133-200: the first node absorbs the others (offsets united, edges moved, equal edges merge their paths)"""
        ref = mns[0]
        merged = {}
        for node in mns:
            merged.update(self.offsets[node])
        self.offsets[ref] = merged
        self.aligned[ref] = 1
        for mn in mns[1:]:
            for (u, a, b), p in list(self.pred[mn].items()):
                self.add_edge(u, ref, p, a, b)
            for (v, a, b), p in list(self.succ[mn].items()):
                self.add_edge(ref, v, p, a, b)
            self.remove_node(mn)
        return ref

    def _bfs(self, source, reverse=False, ignore=()):
        """rem.py:228-258: walk from `source` over edges carried by at least one real (non-'*') path; unaligned nodes are
        walked through (kind 0), aligned nodes stop the walk (kind 1) unless listed in `ignore`, sentinels stop it (kind 2)"""
        adj = self.pred if reverse else self.succ
        aligned = self.aligned
        star = any(p.startswith("*") for p in self.paths)      # (no '*' path at all: every edge is carried by a real one)
        visited = {source}
        queue = [source]
        qi = 0
        while qi < len(queue):
            parent = queue[qi]
            qi += 1
            for key, p in adj[parent].items():
                child = key[0]
                if child in visited or (star and not self._real(p)):
                    continue
                visited.add(child)
                al = aligned.get(child)
                if al is None:
                    yield child, 2
                elif al == 0 or child in ignore:
                    queue.append(child)
                    yield child, 0
                else:
                    yield child, 1

    def segmentgraph(self, node, nodes):
        """rem.py:260-316: of the sub-index' intervals `nodes`, those behind the merged node (trailing: reachable forward
        through unaligned nodes), those in front of it (leading, mirrored), and the rest.

        The reference follows each walk, when it ended at more than one place, by a walk back from every end point and
        intersects (rem.py:282-287, 303-308).  That second step never removes anything: a node of the forward walk goes on,
        through unaligned nodes, to the first aligned node or sentinel on its way -- which the forward walk also reached and
        listed as an end point -- so the walk back from that end point finds it.  It is left out here (it tripled the cost of
        a call, and a merge of graphs spends nearly all its time in these walks); `segmentgraph_literal` keeps the reference's
        form, tests/test_cpu_graphrem.py runs both on every call of the fixture alignments."""
        if self.literal_segments:
            return self.segmentgraph_literal(node, nodes)
        nodes = set(nodes)
        trailing = {c for c, t in self._bfs(node) if t == 0 and isinstance(c, tuple)} & nodes
        leading = {c for c, t in self._bfs(node, reverse=True) if t == 0 and isinstance(c, tuple)} & nodes
        return leading, trailing, nodes - (leading | trailing)

    def check_segment_shortcut(self):
        """the shortcut of segmentgraph needs every sequence node to go on, over an edge carried by a real path, in both directions
        (then a walk through unaligned nodes always ends at an aligned node or a sentinel it lists).  The FASTA reader builds graphs
        like that; a graph read from a file may not (a node no path leaves): it gets the reference's literal form"""
        for n in self.offsets:
            if isinstance(n, tuple):
                if not any(self._real(p) for p in self.succ[n].values()) or not any(self._real(p) for p in self.pred[n].values()):
                    self.literal_segments = True
                    return False
        return True

    def segmentgraph_literal(self, node, nodes):
        """rem.py:260-316 step by step (with the walks back from the end points)"""
        nodes = set(nodes)

        def side(reverse):
            walk, endpoints = set(), set()
            for c, t in self._bfs(node, reverse=reverse):
                if t == 0:
                    walk.add(c)
                else:
                    endpoints.add(c)
            if len(endpoints) > 1:
                back = set()
                for ep in endpoints:
                    for c, t in self._bfs(ep, reverse=not reverse, ignore=endpoints):
                        if t == 0:
                            back.add(c)
                walk &= back
            return {c for c in walk if isinstance(c, tuple)} & nodes
        trailing = side(False)
        leading = side(True)
        return leading, trailing, nodes - (leading | trailing)

    def prune_nodes(self, T):
        """rem.py:384-447: sibling nodes with identical sequence that hang on one parent (or one child) and have no other
        parent (child) are merged, until nothing changes.  The reference rescans every node of the graph after each pass that
        merged something -- merges cascade one step per pass along a region, so a graph of 2*10^5 nodes took minutes -- here a
        merge puts the nodes whose neighbourhood it changed back on a work list; full passes repeat until one merges nothing,
        so the result is the same fixpoint."""
        from collections import deque

        def seq_of(n):
            return self.seq[n] if n in self.seq else (T[n[0]:n[1]] if isinstance(n, tuple) else None)
        while True:
            merged_any = False
            queue = deque(self.offsets)
            queued = set(queue)
            while queue:
                node = queue.popleft()
                queued.discard(node)
                if node not in self.offsets:
                    continue
                for adj, back in ((self.succ, self.pred), (self.pred, self.succ)):
                    neis = [v for (v, a, b) in adj[node] if a == "+" and b == "+"]
                    if len(neis) < 2:
                        continue
                    groups = {}
                    for nei in neis:
                        sq = seq_of(nei)
                        if sq is None:
                            continue
                        groups.setdefault(sq, []).append(nei)
                    for group in groups.values():
                        if len(group) > 1 and all(sum(1 for (u, a, b) in back[v] if a == "+" and b == "+") <= 1 for v in group):
                            ref = self.mergenodes(list(group))
                            merged_any = True
                            again = [node, ref] + [v for (v, a, b) in self.succ[ref]] + [u for (u, a, b) in self.pred[ref]]
                            for x in again:
                                if x not in queued and x in self.offsets:
                                    queue.append(x)
                                    queued.add(x)
            if not merged_any:
                break

    # ---- the graph of a finished run, built behind the ABI ----------------------------------------------------------
    def replay_native(self, root_nodes, an_l, an_off, an_pos):
        """rv_graph_replay (include/reveal_amd.h): graphalign's surgery (rem.py:331-345) for every anchor of a finished run, in C++, the result
        loaded into this graph -- which must be the FASTA reader's (one sequence per sample, nothing aligned yet).  Same nodes, links and path sets
        in the same dictionary order as rem.replay_anchors leaves them, so prune_nodes and write_gfa give the same file.
        root_nodes: the sequences' intervals in sample order; an_l / an_off / an_pos: the anchors as index.align_builtin returns them."""
        with NativeGraph(self, root_nodes, an_l, an_off, an_pos) as ng:
            return ng.load_into(self)

    # ---- invariants used by the tests ----------------------------------------------------------------
    def spell(self, sample, T):
        """the sequence a path spells: walk its edges from its start sentinel (what `reveal extract` prints, test15)"""
        sid = self.path2id[sample]
        out = []
        for start in self.startnodes:
            if start in self.offsets and sid in self.offsets[start]:
                node = start
                while True:
                    nxt = [(v, b) for (v, a, b), p in self.succ[node].items() if sid in p]
                    if len(nxt) != 1:
                        if len(nxt) > 1:
                            raise ValueError("path %s is ambiguous at %s" % (sample, node))
                        break
                    node, strand = nxt[0]
                    if isinstance(node, tuple):
                        if strand != "+":
                            raise NotImplementedError("reverse-strand traversal")
                        out.append(T[node[0]:node[1]])
                    elif node in self.endnodes:
                        break
                break
        return "".join(out).upper()


    def spell_by_offsets(self, sample, T):
        """the same for a graph without sentinels (rem.align removes them, rem.py:708-710): the path's nodes in offset order"""
        sid = self.path2id[sample]
        nodes = sorted((o[sid], n) for n, o in self.offsets.items() if isinstance(n, tuple) and sid in o)
        return "".join(T[b:e] for _, (b, e) in nodes).upper()


class NativeGraph:
    """rv_graph (include/reveal_amd.h): the alignment graph of a finished run with one sequence per sample, kept behind the ABI: the anchors'
    surgery (rv_graph_replay), prune_nodes (rv_graph_prune), the GFA text (rv_graph_gfa) -- and, when somebody wants to look at it, the same graph as
    an AlnGraph (load_into).  `G` is the FASTA reader's graph of the inputs: it lends its path names and sentinels."""

    def __init__(self, G, root_nodes, an_l=None, an_off=None, an_pos=None, sa64=False):
        """with anchors: the graph after their surgery (rv_graph_replay); without: the graph of the sequences alone (rv_graph_replay_begin), for
        index.set_replay_graph -- the run that chooses the anchors then applies them level by level on a thread of its own"""
        import numpy as np
        from . import _lib
        self._g = None
        k = len(root_nodes)
        if len(G.startnodes) != k or len(G.endnodes) != k or any(G.aligned.get(tuple(n)) != 0 for n in root_nodes) or len(G.aligned) != k:
            raise ValueError("NativeGraph: the graph is not the FASTA reader's graph of these sequences")
        if [G.path2id[p] for p in G.paths] != list(range(k)):
            raise ValueError("NativeGraph: path ids are not the samples 0..k-1")
        self._dll = _lib.get(bool(sa64)).dll
        self.names = list(G.paths)
        rb = np.ascontiguousarray([n[0] for n in root_nodes], dtype=np.int64); re_ = np.ascontiguousarray([n[1] for n in root_nodes], dtype=np.int64)
        if an_l is None:
            self._g = self._dll.rv_graph_replay_begin(k, rb.ctypes.data, re_.ctypes.data)
            if not self._g:
                raise MemoryError("rv_graph_replay_begin")
            return
        an_l = np.ascontiguousarray(an_l, dtype=np.uint32); an_off = np.ascontiguousarray(an_off, dtype=np.int64); an_pos = np.ascontiguousarray(an_pos, dtype=np.int64)
        self._g = self._dll.rv_graph_replay(k, rb.ctypes.data, re_.ctypes.data, len(an_l), an_l.ctypes.data, an_off.ctypes.data, an_pos.ctypes.data)
        if not self._g:
            raise MemoryError("rv_graph_replay")
        why = self._dll.rv_graph_error(self._g)
        if why:
            self.close()
            raise RuntimeError(why.decode())

    def close(self):
        if getattr(self, "_g", None):
            self._dll.rv_graph_free(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    @staticmethod
    def _text(T):
        return T if isinstance(T, (bytes, bytearray)) else T.encode("latin-1")

    def counts(self):
        """-> (sequence nodes, links) as the graph stands"""
        import ctypes
        import numpy as np
        sz = np.zeros(4, dtype=np.int64)
        self._dll.rv_graph_sizes(self._g, sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        return int(sz[0]) - 2 * len(self.names), int(sz[2])

    def prune(self, T):
        """AlnGraph.prune_nodes (rem.py:384-447) on the graph behind the ABI; T = the index text after the run"""
        self._dll.rv_graph_prune(self._g, self._text(T))

    def gfa(self, T, cmdline=None):
        """the text write_gfa writes -> bytes"""
        import ctypes
        names = (ctypes.c_char_p * len(self.names))(*[n.encode() for n in self.names])
        out = ctypes.c_char_p()
        cl = (cmdline if cmdline is not None else " ".join(sys.argv)).encode()
        n = self._dll.rv_graph_gfa(self._g, self._text(T), len(self.names), names, cl, ctypes.byref(out))
        return ctypes.string_at(out, n)

    def write_gfa(self, T, outputfile, cmdline=None):
        """write_gfa(G, T, outputfile) for the graph behind the ABI -> the file name written"""
        if not outputfile.endswith(".gfa") and not outputfile.endswith(".gfa.gz"):
            outputfile += ".gfa.gz"
        data = self.gfa(T, cmdline)
        with (gzip.open if outputfile.endswith(".gz") else open)(outputfile, "wb") as f:
            f.write(data)
        return outputfile

    def load_into(self, G):
        """the graph as it stands behind the ABI into the AlnGraph `G` (the one handed to the constructor): nodes, links and path sets in the
        dictionary order the Python surgery leaves them in -> number of nodes"""
        import ctypes
        import numpy as np
        dll, g = self._dll, self._g
        k = len(self.names)
        sz = np.zeros(4, dtype=np.int64)
        dll.rv_graph_sizes(g, sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        nn, no, ne, npth = (int(x) for x in sz)
        nb = np.zeros(nn, np.int64); nend = np.zeros(nn, np.int64); nal = np.zeros(nn, np.int8)
        optr = np.zeros(nn + 1, np.int64); osid = np.zeros(max(no, 1), np.int32); oval = np.zeros(max(no, 1), np.int64)
        sptr = np.zeros(nn + 1, np.int64); sto = np.zeros(max(ne, 1), np.int32); sed = np.zeros(max(ne, 1), np.int32)
        pptr = np.zeros(nn + 1, np.int64); pfr = np.zeros(max(ne, 1), np.int32); ped = np.zeros(max(ne, 1), np.int32)
        eptr = np.zeros(ne + 1, np.int64); epath = np.zeros(max(npth, 1), np.int32)
        dll.rv_graph_export(g, *(x.ctypes.data for x in (nb, nend, nal, optr, osid, oval, sptr, sto, sed, pptr, pfr, ped, eptr, epath)))
        nbl, nel, nall = nb.tolist(), nend.tolist(), nal.tolist()
        names = self._node_names(G, nbl, nel, nall)
        osidl, ovall, optrl = osid.tolist(), oval.tolist(), optr.tolist()
        eptrl, epl = eptr.tolist(), epath.tolist()
        sets = [set(epl[eptrl[j]:eptrl[j + 1]]) for j in range(ne)]      # one object per edge, shared by its two directories (add_edge unites in place)
        seq_keep = {n: G.seq[n] for n in G.seq if not isinstance(n, tuple)}
        offsets, aligned, succ, pred = {}, {}, {}, {}
        sptrl, stol, sedl, pptrl, pfrl, pedl = sptr.tolist(), sto.tolist(), sed.tolist(), pptr.tolist(), pfr.tolist(), ped.tolist()
        # (whole lists at once where a loop body per node is not needed: the keys of all links, the pairs of all offsets)
        skeys = [(names[v], "+", "+") for v in stol[:ne]]
        pkeys = [(names[u], "+", "+") for u in pfrl[:ne]]
        ssets = [sets[e] for e in sedl[:ne]]
        psets = [sets[e] for e in pedl[:ne]]
        opairs = list(zip(osidl[:no], ovall[:no]))
        for i, name in enumerate(names):
            a, b = optrl[i], optrl[i + 1]
            offsets[name] = dict(opairs[a:b])
            a, b = sptrl[i], sptrl[i + 1]
            succ[name] = dict(zip(skeys[a:b], ssets[a:b]))
            a, b = pptrl[i], pptrl[i + 1]
            pred[name] = dict(zip(pkeys[a:b], psets[a:b]))
        aligned = {name: al for name, al in zip(names, nall) if al >= 0}
        G.offsets, G.aligned, G.succ, G.pred, G.seq = offsets, aligned, succ, pred, seq_keep
        G._begins = _Begins(b for b, al in zip(nbl, nall) if al >= 0)
        G._end_of = {b: e for b, e, al in zip(nbl, nel, nall) if al >= 0}
        return nn


    def _node_names(self, G, nbl, nel, nall):
        # sentinels keep their names: (sample, 0 / 1) -> the reader's start / end node of that sequence
        k = len(self.names)
        sid_of = {}
        for st in G.startnodes:
            (sid,) = G.offsets[st].keys(); sid_of[(sid, 0)] = st
        for en in G.endnodes:
            (sid,) = G.offsets[en].keys(); sid_of[(sid, 1)] = en
        if sorted(q for q, _ in sid_of) != sorted(list(range(k)) * 2):
            raise ValueError("NativeGraph: path ids are not the samples 0..k-1")
        return [(sid_of[(b, e)] if al < 0 else (b, e)) for b, e, al in zip(nbl, nel, nall)]


def graph_arrays(G):
    """the graph as the readers left it, in the arrays rv_graph_import takes (include/reveal_amd.h) -> (dict of numpy arrays, node list in dictionary order).
    Raises ValueError for a link on the reverse strand (such inputs keep the Python callbacks)."""
    import numpy as np
    nodes = list(G.offsets)
    num = {n: i for i, n in enumerate(nodes)}
    nn = len(nodes)
    startset, endset = set(G.startnodes), set(G.endnodes)
    nb = np.zeros(nn, np.int64); ne = np.zeros(nn, np.int64); al = np.zeros(nn, np.int8); sent = np.zeros(nn, np.int8)
    optr = np.zeros(nn + 1, np.int64)
    osid, oval = [], []
    aligned = G.aligned
    for i, n in enumerate(nodes):
        if isinstance(n, tuple):
            nb[i], ne[i] = n
            al[i] = aligned.get(n, 0)
        else:
            nb[i], ne[i], al[i] = i, 0, -1
            sent[i] = 1 if n in startset else 2 if n in endset else 0
        o = G.offsets[n]
        osid.extend(o.keys()); oval.extend(o.values())
        optr[i + 1] = len(osid)
    eu, ev, eptr, epaths = [], [], [0], []
    eid = {}
    for n in nodes:
        u = num[n]
        for (v, a, b), p in G.succ[n].items():
            if a != "+" or b != "+":
                raise ValueError("a link on the reverse strand: %s -> %s" % (n, v))
            eid[(u, num[v])] = len(eu)
            eu.append(u); ev.append(num[v])
            epaths.extend(sorted(p)); eptr.append(len(epaths))
    pptr = np.zeros(nn + 1, np.int64)
    pedge = []
    for i, n in enumerate(nodes):
        for (u, a, b) in G.pred[n]:
            pedge.append(eid[(num[u], i)])
        pptr[i + 1] = len(pedge)
    npaths = len(G.paths)
    star = np.array([1 if G.id2path[sid].startswith("*") else 0 for sid in range(npaths)], np.uint8)
    id2end = np.array([G.id2end.get(sid, 0) for sid in range(npaths)], np.int64)
    starts = np.array([num[x] for x in G.startnodes if x in num], np.int32)
    arr = dict(nb=nb, ne=ne, al=al, sent=sent, optr=optr, osid=np.array(osid or [0], np.int32), oval=np.array(oval or [0], np.int64),
               eu=np.array(eu or [0], np.int32), ev=np.array(ev or [0], np.int32), eptr=np.array(eptr, np.int64), epaths=np.array(epaths or [0], np.int32),
               pptr=pptr, pedge=np.array(pedge or [0], np.int32), star=star, id2end=id2end, starts=starts, nedges=len(eu), npaths=npaths,
               literal=1 if G.literal_segments else 0)
    return arr, nodes


class ReverseStrand(ValueError):
    """a GFA input with links on the reverse strand: not read behind the ABI"""


class LoopGraph(NativeGraph):
    """rv_graph behind the ABI for GRAPH inputs (include/reveal_amd.h rv_graph_import): the structure the library's own picker and graphalign work on
    while the recursion runs (rv_set_graph_picker), made from the AlnGraph the readers left.  pick / align expose the two steps one call at a time (the tests run
    them beside schemes.GraphPicker / rem.GraphAligner on every sub-index of whole alignments)."""

    def __init__(self, G, sa64=False):
        from . import _lib
        self._lib = _lib.get(bool(sa64))
        self._dll = self._lib.dll
        self.names = list(G.paths)
        a, self.nodes0 = graph_arrays(G)
        p = lambda x: x.ctypes.data
        self._g = self._dll.rv_graph_import(len(a["nb"]), p(a["nb"]), p(a["ne"]), p(a["al"]), p(a["sent"]), p(a["optr"]), p(a["osid"]), p(a["oval"]),
                                            a["nedges"], p(a["eu"]), p(a["ev"]), p(a["eptr"]), p(a["epaths"]), p(a["pptr"]), p(a["pedge"]), a["npaths"], p(a["star"]), p(a["id2end"]),
                                            len(a["starts"]), p(a["starts"]), a["literal"])
        if not self._g:
            raise RuntimeError(self._lib.err())
        self.sentinels0 = [n for n in self.nodes0 if not isinstance(n, tuple)]

    @classmethod
    def read(cls, inputfiles, idx, G, contigs=True, toupper=True, sa64=False, threads=None):
        """the inputs read behind the ABI (rv_graph_add_linear / rv_graph_read_gfa: csrc/rv_gfaread.hip): sequences go to `idx` (reveal_amd's index) as the
        Python readers would add them, the graph is made where the run uses it; of `G` only the path tables are filled.  Raises ReverseStrand when a file
        holds links on the reverse strand -- idx and G are half-filled then: start over with the Python readers."""
        import ctypes
        import uuid
        from . import _lib
        from .rem import fasta_reader
        self = cls.__new__(cls)
        self._lib = _lib.get(bool(sa64))
        self._dll = dll = self._lib.dll
        self._g = dll.rv_graph_new()
        if not self._g:
            raise MemoryError(self._lib.err())
        if getattr(idx, "_h", None) is not None:
            # one allocation of the (page-locked) host text: a file is never shorter than the text it adds
            total = sum(os.path.getsize(f) for f in inputfiles if not f.endswith(".gz"))
            dll.rv_reserve_text(idx._h, int(dll.rv_n(idx._h)) + total + 64)
        # graph files are parsed side by side (rv_gfa_parse touches neither graph nor index; the library call runs without the interpreter's lock) and adopted in
        # the order of the inputs (rv_graph_adopt)
        gfas = [f for f in inputfiles if f.endswith(".gfa") or f.endswith(".gfa.gz")]

        def parse(f):
            with (gzip.open if f.endswith(".gz") else open)(f, "rb") as fh:
                data = fh.read()
            return dll.rv_gfa_parse(data, len(data))
        pool, pending = None, {}
        if len(gfas) > 1 and threads != 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=min(len(gfas), threads or 4))
            pending = {f: pool.submit(parse, f) for f in gfas}
        for f in inputfiles:
            if f.endswith(".gfa") or f.endswith(".gfa.gz"):
                idx.addsample(os.path.basename(f))
                parsed = pending.pop(f).result() if f in pending else parse(f)
                if not parsed:
                    self.close()
                    raise MemoryError(self._lib.err())
                names = ctypes.c_char_p()
                if getattr(idx, "_h", None) is not None:
                    n = dll.rv_graph_adopt(self._g, idx._h, None, parsed, ctypes.byref(names))
                else:      # (an index stand-in that only counts text positions -- tests without a device: the graph alone)
                    tn = ctypes.c_int64(idx.n)
                    n = dll.rv_graph_adopt(self._g, None, ctypes.byref(tn), parsed, ctypes.byref(names))
                    idx.n = tn.value
                why = self._lib.err() if n == -1 else None
                dll.rv_gfa_parsed_free(parsed)
                if n < 0:
                    for fut in pending.values():
                        p2 = fut.result()
                        if p2:
                            dll.rv_gfa_parsed_free(p2)
                    pending = {}
                    if pool is not None:
                        pool.shutdown()
                if n == -2:
                    self.close()
                    raise ReverseStrand(f)
                if n < 0:
                    self.close()
                    raise ValueError("%s: %s" % (f, why))
                for name in names.value.decode("latin-1").split("\n")[:n]:
                    _new_path(G, name)
            else:
                if contigs:
                    idx.addsample(os.path.basename(f))
                for name, seq in fasta_reader(f, toupper=toupper):
                    if not contigs:
                        idx.addsample(name)
                    name = name.replace(":", "").replace(";", "")
                    sid = _new_path(G, name, len(seq))
                    b, e = idx.addsequence(seq)
                    if dll.rv_graph_add_linear(self._g, b, e, 1 if name.startswith("*") else 0) != sid:
                        raise RuntimeError("path ids out of step: " + self._lib.err())
        if pool is not None:
            pool.shutdown()
        if dll.rv_graph_seal(self._g) != 0:
            raise MemoryError(self._lib.err())
        if hasattr(idx, "_sync_nodes"):
            idx._nodes_stale = True      # (idx.nodes asks the library when somebody looks)
            idx._constructed = False
        k = dll.rv_graph_paths(self._g, None)
        import numpy as np
        ends = np.zeros(max(k, 1), np.int64)
        dll.rv_graph_paths(self._g, ends.ctypes.data)
        G.id2end.update({sid: int(ends[sid]) for sid in range(k)})
        self.names = list(G.paths)
        self.nodes0, self.sentinels0 = None, None
        return self

    def _sentinel_names(self, G):
        """names for the graph's sentinels when it was read behind the ABI (they have none there), registered with G as the Python readers would"""
        import ctypes
        import uuid
        import numpy as np
        sz = np.zeros(4, dtype=np.int64)
        self._dll.rv_graph_sizes(self._g, sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        kinds = np.zeros(max(int(sz[0]), 1), np.int8)
        self._dll.rv_graph_node_kinds(self._g, kinds.ctypes.data)
        self.sentinels0 = []
        G.startnodes, G.endnodes = [], []
        for kd in kinds[:int(sz[0])].tolist():
            if kd:
                name = uuid.uuid4().hex
                self.sentinels0.append(name)
                (G.startnodes if kd == 1 else G.endnodes).append(name)

    def finish(self):
        """after the run (index.set_graph_picker + align_builtin): live nodes and links renumbered, ready for counts / prune / gfa / load_into"""
        self._dll.rv_graph_finish(self._g)

    def counts(self):
        import ctypes
        import numpy as np
        sz = np.zeros(4, dtype=np.int64)
        self._dll.rv_graph_sizes(self._g, sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        kinds = np.zeros(max(int(sz[0]), 1), np.int8)
        self._dll.rv_graph_node_kinds(self._g, kinds.ctypes.data)
        return int(sz[0]) - int(np.count_nonzero(kinds[:int(sz[0])])), int(sz[2])

    def _node_names(self, G, nbl, nel, nall):
        # sentinels are neither made nor removed by the run and keep their order: the k-th one is the k-th one of the graph handed to the constructor
        if self.sentinels0 is None:
            self._sentinel_names(G)
        it = iter(self.sentinels0)
        return [(next(it) if al < 0 else (b, e)) for b, e, al in zip(nbl, nel, nall)]

    @staticmethod
    def _iv(node):
        import numpy as np
        return np.array([-1, -1] if node is None else [node[0], node[1]], np.int64)

    def pick(self, mums, nsub, leftnode, rightnode, args, minlength=20):
        """schemes.GraphPicker.graphmumpicker(mums, idx, precomputed=False) -> () or (mum, skipleft, skipright)"""
        import ctypes
        import numpy as np
        from . import schemes
        m = len(mums)
        members = sum(len(mm[2]) for mm in mums)
        ln = np.ascontiguousarray([mm[0] for mm in mums], dtype=np.uint32); nn = np.ascontiguousarray([mm[1] for mm in mums], dtype=np.int32)
        off = np.zeros(m + 1, np.int64); so = np.zeros(max(members, 1), np.uint16); pos = np.zeros(max(members, 1), np.int64)
        w = 0
        for i, mm in enumerate(mums):
            for gq, pq in mm[2]:
                so[w] = gq; pos[w] = pq; w += 1
            off[i + 1] = w
        ns = max((len(mm[2]) for mm in mums), default=1)
        A = schemes._RvPickerArgs(int(args.wscore), int(args.wpen), int(args.maxmums or 0), int(args.seedsize or 0), schemes.GCMODELS[args.gcmodel], 1 if args.trim else 0, float(args.pcutoff))
        pso = np.zeros(ns, np.uint16); ppos = np.zeros(ns, np.int64)
        cap = max(m, 1)
        sl = np.zeros(cap, np.uint32); sn = np.zeros(cap, np.int32); soff = np.zeros(cap + 1, np.int64)
        sso = np.zeros(max(members, 1), np.uint16); spos = np.zeros(max(members, 1), np.int64); ssc = np.zeros(cap, np.int64); srt = np.zeros(cap, np.uint8)
        O = schemes._RvPickerOut()
        O.pick_so, O.pick_pos, O.member_cap = pso.ctypes.data, ppos.ctypes.data, ns
        O.seed_cap, O.seed_member_cap = cap, max(members, 1)
        O.seed_l, O.seed_n, O.seed_off, O.seed_so, O.seed_pos, O.seed_score, O.seed_right = (x.ctypes.data for x in (sl, sn, soff, sso, spos, ssc, srt))
        lf, rt = self._iv(leftnode), self._iv(rightnode)
        r = self._dll.rv_graph_pick(self._g, ctypes.byref(A), int(nsub), m, ln.ctypes.data, nn.ctypes.data, off.ctypes.data, so.ctypes.data, pos.ctypes.data,
                                    lf.ctypes.data, rt.ctypes.data, int(minlength), ctypes.byref(O))
        if r < 0:
            raise RuntimeError(self._lib.err())
        if r == 0:
            return ()
        pick = (int(O.pick_l), int(O.pick_n), tuple((int(pso[q]), int(ppos[q])) for q in range(O.pick_members)))
        left, right = [], []
        for k in range(O.nleft + O.nright):
            mm = (int(sl[k]), int(sn[k]), tuple((int(sso[q]), int(spos[q])) for q in range(soff[k], soff[k + 1])))
            (right if srt[k] else left).append((mm, int(ssc[k])))
        return pick, left, right

    def align(self, nodes, leftnode, rightnode, mum):
        """rem.GraphAligner.graphalign -> (leading, trailing, matching, rest, merged, newleft, newright) with the interval collections as sorted lists"""
        import numpy as np
        l, n, spd = mum
        nd = np.ascontiguousarray(sorted(nodes), dtype=np.int64).reshape(-1, 2)
        ps = np.ascontiguousarray([p for _, p in spd], dtype=np.int64)
        counts = np.zeros(4, np.int64); out6 = np.zeros(6, np.int64)
        lf, rt = self._iv(leftnode), self._iv(rightnode)
        if self._dll.rv_graph_align(self._g, nd.ctypes.data, len(nd), lf.ctypes.data, rt.ctypes.data, int(l), ps.ctypes.data, len(ps), counts.ctypes.data, out6.ctypes.data) != 0:
            raise RuntimeError(self._lib.err())
        buf = np.zeros(2 * max(int(counts.sum()), 1), np.int64)
        self._dll.rv_graph_align_fetch(self._g, buf.ctypes.data)
        ivs = [tuple(x) for x in buf[:2 * int(counts.sum())].reshape(-1, 2).tolist()]
        c = [int(x) for x in counts]
        lead, trail, match, rest = ivs[:c[0]], ivs[c[0]:c[0] + c[1]], ivs[c[0] + c[1]:c[0] + c[1] + c[2]], ivs[c[0] + c[1] + c[2]:]
        node = lambda b, e: None if b < 0 else (int(b), int(e))
        return lead, trail, match, rest, node(out6[0], out6[1]), node(out6[2], out6[3]), node(out6[4], out6[5])

    def snapshot(self):
        """-> (nodes, offsets, links): sequence nodes as (b, e, aligned), sentinels as ('s', k) in order of appearance; per node its (path, offset) list and its
        links forwards as (target, sorted path ids), all in dictionary order"""
        import ctypes
        import numpy as np
        dll, g = self._dll, self._g
        dll.rv_graph_finish(g)
        sz = np.zeros(4, np.int64)
        dll.rv_graph_sizes(g, sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        nn, no, ne, npth = (int(x) for x in sz)
        nb = np.zeros(nn, np.int64); nend = np.zeros(nn, np.int64); nal = np.zeros(nn, np.int8)
        optr = np.zeros(nn + 1, np.int64); osid = np.zeros(max(no, 1), np.int32); oval = np.zeros(max(no, 1), np.int64)
        sptr = np.zeros(nn + 1, np.int64); sto = np.zeros(max(ne, 1), np.int32); sed = np.zeros(max(ne, 1), np.int32)
        pptr = np.zeros(nn + 1, np.int64); pfr = np.zeros(max(ne, 1), np.int32); ped = np.zeros(max(ne, 1), np.int32)
        eptr = np.zeros(ne + 1, np.int64); epath = np.zeros(max(npth, 1), np.int32)
        dll.rv_graph_export(g, *(x.ctypes.data for x in (nb, nend, nal, optr, osid, oval, sptr, sto, sed, pptr, pfr, ped, eptr, epath)))
        names, ks = [], 0
        for i in range(nn):
            if nal[i] < 0:
                names.append(("s", ks)); ks += 1
            else:
                names.append((int(nb[i]), int(nend[i]), int(nal[i])))
        offs = [list(zip(osid[optr[i]:optr[i + 1]].tolist(), oval[optr[i]:optr[i + 1]].tolist())) for i in range(nn)]
        links = [[(names[sto[q]], epath[eptr[sed[q]]:eptr[sed[q] + 1]].tolist()) for q in range(sptr[i], sptr[i + 1])] for i in range(nn)]
        preds = [[names[pfr[q]] for q in range(pptr[i], pptr[i + 1])] for i in range(nn)]
        return names, offs, links, preds


def graph_snapshot(G):
    """the same view of a Python AlnGraph (LoopGraph.snapshot)"""
    names, ks = {}, 0
    for n in G.offsets:
        if isinstance(n, tuple):
            names[n] = (n[0], n[1], G.aligned.get(n, 0))
        else:
            names[n] = ("s", ks); ks += 1
    order = list(G.offsets)
    offs = [list(G.offsets[n].items()) for n in order]
    links = [[(names[v], sorted(p)) for (v, a, b), p in G.succ[n].items()] for n in order]
    preds = [[names[u] for (u, a, b) in G.pred[n]] for n in order]
    return [names[n] for n in order], offs, links, preds


# ---- readers (reveal/utils.py:304-375, 377-677) -------------------------------------------------------

def _new_path(G, name, length=None):
    if name in G.path2id:
        raise ValueError("the graph already contains a path named %r" % name)
    sid = len(G.path2id)
    G.paths.append(name)
    G.path2id[name] = sid
    G.id2path[sid] = name
    if length is not None:
        G.id2end[sid] = length
    return sid


def read_fasta(fasta, index, G, contigs=True, toupper=True):
    """utils.py:304-375: one sample per file (or per sequence with contigs=False), one node + start / end sentinel per sequence"""
    from .rem import fasta_reader
    import os
    if contigs:
        index.addsample(os.path.basename(fasta))
    for name, seq in fasta_reader(fasta, toupper=toupper):
        if not contigs:
            index.addsample(name)
        name = name.replace(":", "").replace(";", "")
        sid = _new_path(G, name, len(seq))
        intv = tuple(index.addsequence(seq))
        start, end = uuid.uuid4().hex, uuid.uuid4().hex
        G.add_node(start, offsets={sid: 0})
        G.startnodes.append(start)
        G.add_node(intv, offsets={sid: 0}, aligned=0)
        G.add_node(end, offsets={sid: len(seq)})
        G.endnodes.append(end)
        G.add_edge(start, intv, {sid})
        G.add_edge(intv, end, {sid})


def read_gfa(gfafile, index, G):
    """utils.py:377-677 (the defaults `reveal rem` uses for graph inputs: every S-line becomes one '$'-terminated sequence of
    the current sample; P-lines give the node offsets; untraversed nodes / edges go; per connected component the start
    sentinels of its paths are merged into one, likewise the end sentinels)"""
    fopen = gzip.open if gfafile.endswith(".gz") else open
    nmap, edges, plines = {}, [], []
    with fopen(gfafile, "rt") as f:
        for line in f:
            if line.startswith("S"):
                s = line.rstrip("\n").split("\t")
                seq = s[2] if len(s) > 2 else ""
                intv = tuple(index.addsequence(seq.upper()))
                G.add_node(intv, offsets={}, aligned=0)
                nmap[s[1]] = intv
            elif line.startswith("L"):
                edges.append(line)
            elif line.startswith("P"):
                plines.append(line)
    for line in edges:
        e = line.rstrip("\n").split("\t")
        G.add_edge(nmap[e[1]], nmap[e[3]], set(), e[2], e[4])
    if not plines:
        raise ValueError("no paths defined in %s" % gfafile)
    starts, ends = {}, {}          # (dictionaries for their order: the reference keeps sets of random names and meets them in any order; here creation order)
    for line in plines:
        cols = line.rstrip("\n").split("\t")
        sample = cols[1]
        sid = _new_path(G, sample)
        o = 0
        path = [(x[:-1], x[-1:]) for x in cols[2].split(",")] if len(cols) >= 3 and cols[2] else []
        prev = None
        for nid, strand in path:
            node = nmap[nid]
            G.offsets[node][sid] = o
            o += node[1] - node[0]
            if prev is not None:
                key = (node, prev[1], strand)
                if key not in G.succ[prev[0]]:
                    raise ValueError("path %s steps %s -> %s but the graph has no such link" % (sample, prev[0], node))
                G.succ[prev[0]][key].add(sid)
            prev = (node, strand)
        start, end = uuid.uuid4().hex, uuid.uuid4().hex
        G.add_node(start, offsets={sid: 0})
        G.add_node(end, offsets={sid: o})
        if path:
            G.add_edge(start, nmap[path[0][0]], {sid}, "+", path[0][1])
            G.add_edge(nmap[path[-1][0]], end, {sid}, path[-1][1], "+")
        starts[start] = None; ends[end] = None
        G.id2end[sid] = o
    for u in list(G.succ):                              # untraversed edges, then untraversed nodes
        for key, p in list(G.succ[u].items()):
            if not p:
                del G.succ[u][key]
                del G.pred[key[0]][(u, key[1], key[2])]
    mine = set(nmap.values())
    for n in [n for n in mine if not G.offsets[n]]:
        G.remove_node(n)
        mine.discard(n)
    # weakly connected components of what this file added -- each found from its first member in creation order --; one start and one end sentinel per
    # component, made from the paths' own in creation order (rv_gfaread.hip does the same)
    todo = mine | set(starts) | set(ends)
    comps = []
    for x0 in [n for n in G.offsets if n in todo]:
        if x0 not in todo:
            continue
        comp, stack = set(), [x0]
        while stack:
            x = stack.pop()
            if x in comp:
                continue
            comp.add(x)
            stack.extend(v for (v, a, b) in G.succ[x] if v not in comp)
            stack.extend(u for (u, a, b) in G.pred[x] if u not in comp)
        todo -= comp
        comps.append(comp)
    for comp in comps:
        for group, register, forward in (([x for x in ends if x in comp], G.endnodes, False), ([x for x in starts if x in comp], G.startnodes, True)):
            if not group:
                continue
            sentinel = uuid.uuid4().hex
            G.add_node(sentinel, offsets={}, seq="")
            register.append(sentinel)
            for old in group:
                G.offsets[sentinel].update(G.offsets[old])
                if forward:
                    for (v, a, b), p in list(G.succ[old].items()):
                        G.add_edge(sentinel, v, p, a, b)
                else:
                    for (u, a, b), p in list(G.pred[old].items()):
                        G.add_edge(u, sentinel, p, a, b)
                G.remove_node(old)
    G.check_segment_shortcut()


# ---- writer (reveal/utils.py:710-839 after rem.align_cmd's seq2node, utils.py:1036-1049) ----------------

def write_gfa(G, T, outputfile, cmdline=None):
    """GFA1: S per sequence node (ids 1.. in node order; aligned nodes upper-cased as seq2node does), L per edge between
    sequence nodes, P per path (walked from the start sentinels).  -> the file name written"""
    if not outputfile.endswith(".gfa") and not outputfile.endswith(".gfa.gz"):
        outputfile += ".gfa.gz"
    fopen = gzip.open if outputfile.endswith(".gz") else open
    nodes = G.seq_nodes()
    ident = {n: i + 1 for i, n in enumerate(nodes)}
    with fopen(outputfile, "wt") as f:
        f.write("H\tVN:Z:1.0\tCL:Z:%s\n" % (cmdline if cmdline is not None else " ".join(sys.argv)))
        for n in nodes:
            s = G.seq[n] if n in G.seq else T[n[0]:n[1]]
            if G.aligned.get(n, 0) > 0:
                s = s.upper()
            f.write("S\t%d\t%s\n" % (ident[n], s))
            for (v, a, b), p in G.succ[n].items():
                if isinstance(v, tuple):
                    f.write("L\t%d\t%s\t%d\t%s\t0M\n" % (ident[n], a, ident[v], b))
        endset = set(G.endnodes)
        for sample, sid in G.path2id.items():
            path, cigar = [], []
            for start in G.startnodes:
                if start not in G.offsets or sid not in G.offsets[start]:
                    continue
                node = start
                while True:
                    out = [(v, b) for (v, a, b), p in G.succ[node].items() if sid in p]
                    if len(out) != 1:
                        break
                    v, strand = out[0]
                    if v in endset:
                        break
                    if isinstance(v, tuple):
                        path.append("%d%s" % (ident[v], strand))
                        if isinstance(node, tuple):
                            cigar.append("0M")
                    node = v
                break
            f.write("P\t%s\t%s\t%s\n" % (sample, ",".join(path), ",".join(cigar)))
    return outputfile
