"""One alignment over several GPUs: the frontier hand-off of include/reveal_amd.h.

SURVEY.md 8(e), second granularity.  The reference's recursion is a stack of
independent sub-indices popped by worker threads (reveallib/reveal.c:21-25,
:731-1338): children of a split cover disjoint text and disjoint ranges of the
shared inverse and are pushed after their parent's lower-casing (:1230-1234
before :1296).  Here the workers are GPUs.  Rank 0 builds the index and runs the
level loop until the frontier is wide enough, every rank (rank 0 included)
receives a share of the sub-indices -- metadata plus their SA / LCP / BWT
segments, 9 B per rank, point to point (RCCL send/recv over xGMI with the nccl
backend, host memory with gloo) -- and finishes it with the same level loop on a
handle that only holds the text.  Anchors are gathered on rank 0; their union is
the anchor set of the undivided run (tests/test_gpu_handoff.py compares it, the
per-sub-index trace and the lower-cased text with the oracle's).

No collective on the data path.  What stays on one GPU (construct + the first
levels) bounds the speed-up (Amdahl): at 2 x 250 Mbp the top of the recursion is
wide (every level streams all 5*10^8 ranks), so dividing pays; for small inputs
independent alignments per GPU (bench.py's default) are the better use of a node.
"""
import heapq

import numpy as np

STAT_SUM = ("steps", "splits", "anchored_bp", "scanned_ranks")
STAT_MAX = ("levels", "maxdepth", "t_scan", "t_host", "t_split", "t_bubble")


def partition(sizes, nparts):
    """sub-indices -> nparts shares, largest first into the lightest share (LPT).
    -> list of ascending int32 arrays (sub-index ids; a share may be empty)"""
    sizes = np.asarray(sizes, dtype=np.int64)
    bins = [[] for _ in range(nparts)]
    heap = [(0, k) for k in range(nparts)]
    for s in np.argsort(-sizes, kind="stable"):
        load, k = heapq.heappop(heap)
        bins[k].append(int(s))
        heapq.heappush(heap, (load + int(sizes[s]), k))
    return [np.asarray(sorted(b), dtype=np.int32) for b in bins]


def subset(fr, subs):
    """the part of a frontier() dict that describes the sub-indices `subs` (in that order)"""
    subs = np.asarray(subs, dtype=np.int64)
    nf = fr["node_first"]
    cnt = nf[subs + 1] - nf[subs] if len(subs) else np.zeros(0, dtype=np.int64)
    first = np.zeros(len(subs) + 1, dtype=np.int64)
    np.cumsum(cnt, out=first[1:])
    if len(subs):
        take = np.concatenate([np.arange(nf[s], nf[s + 1]) for s in subs])
        nodes = fr["nodes"][take]
    else:
        nodes = np.zeros((0, 2), dtype=np.int64)
    return dict(level=fr["level"], meta=fr["meta"][subs].copy(), node_first=first, nodes=nodes)


def empty_result(trace=False):
    from . import _lib
    st = {f[0]: 0 for f in _lib.RvAlignStats._fields_}
    tr = np.zeros(0, dtype=_lib.TRACE_DTYPE) if trace else None
    return dict(stats=st, anchors=(np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros(0, np.int64)), trace=tr)


def merge(results):
    """results of align_builtin_resume() of every share -> one result of the same shape"""
    results = [r for r in results if r is not None]
    l = np.concatenate([r["anchors"][0] for r in results])
    pos = np.concatenate([r["anchors"][2] for r in results])
    cnt = np.concatenate([np.diff(r["anchors"][1]) for r in results])
    off = np.zeros(len(l) + 1, dtype=np.int64)
    np.cumsum(cnt, out=off[1:])
    st = dict(results[0]["stats"])
    for k in STAT_SUM:
        st[k] = sum(r["stats"][k] for r in results)
    for k in STAT_MAX:
        st[k] = max(r["stats"][k] for r in results)
    tr = None
    if all(r["trace"] is not None for r in results):
        tr = np.concatenate([r["trace"] for r in results])
    return dict(stats=st, anchors=(l, off, pos), trace=tr)


def lower_text(T, anchors):
    """the text after the recursion: every member of every anchor lower-cased (reveal.c:1230-1234)"""
    t = np.frombuffer(T, dtype=np.uint8).copy() if isinstance(T, (bytes, bytearray)) else np.array(T, dtype=np.uint8)
    l, off, pos = anchors
    if len(l):
        ll = np.repeat(l.astype(np.int64), np.diff(off))
        d = np.zeros(len(t) + 1, dtype=np.int64)
        np.add.at(d, pos, 1)
        np.add.at(d, pos + ll, -1)
        inside = np.cumsum(d[:-1]) > 0
        up = inside & (t >= 65) & (t <= 90)
        t[up] += 32
    return t


def _buffers(lib, m, device):
    """SA / LCP / BWT transport buffers of m ranks: torch tensors on `device` ('cpu' or 'cuda:k')"""
    import torch
    sa_dt = torch.int64 if lib.sa64 else torch.int32
    return (torch.empty(max(m, 1), dtype=sa_dt, device=device), torch.empty(max(m, 1), dtype=torch.int32, device=device),
            torch.empty(max(m, 1), dtype=torch.uint8, device=device))


def balanced_frontier(idx, world, stop_subs, minl, minn, trace=False, tolerance=1.15, max_subs=None):
    """owner side: levels until the frontier is wide enough AND its largest-first partition is balanced -- a share may exceed
    the mean by `tolerance` at most -- or the frontier has grown to max_subs sub-indices (each further level costs the owner a
    pass over the ranks it still holds, so the widening stops there).  The top of a recursion tree is lopsided (the longest
    match splits a genome anywhere), and a sub-index is the unit of work: the widening is what divides a share that is too
    large for one device.  -> (frontier size, frontier dict or None, parts)"""
    left = idx.align_builtin_until(stop_subs, minl, minn, trace=trace)
    max_subs = max_subs or 64 * world
    while left > 0:
        fr = idx.frontier()
        sizes = fr["meta"][:, 1]
        parts = partition(sizes, world)
        loads = np.array([int(sizes[p].sum()) for p in parts], dtype=np.float64)
        if loads.max() <= tolerance * loads.mean() or left >= max_subs:
            return left, fr, parts
        left = idx.align_builtin_continue(min(2 * left, max_subs))
    return 0, None, [np.zeros(0, np.int32)] * world


def align_sharded(idx, minl=20, minn=2, stop_subs=None, group=None, trace=False):
    """Divide ONE alignment over the ranks of `group`.  Every rank passes an index that holds the same samples
    (addsample / addsequence done, construct not needed); rank 0's is constructed here.
    -> on rank 0 the merged result (shape of index.align_builtin, plus 'shares' = ranks handed to each rank);
       None on the other ranks."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_dev = dist.get_backend(group) == "nccl"
    dev = ("cuda:%d" % torch.cuda.current_device()) if on_dev else "cpu"
    lib = idx._lib
    if world == 1:
        idx.construct()
        res = idx.align_builtin(minl, minn, trace=trace)
        res["shares"] = [int(idx.n)]
        return res
    shares = None
    if rank == 0:
        idx.construct()
        left, fr, parts = balanced_frontier(idx, world, stop_subs or 4 * world, minl, minn, trace=trace)
        if left > 0:
            heads = [dict(part=subset(fr, p), maxlcp=idx.maxlcp) for p in parts]
        else:
            heads = [dict(part=None, maxlcp=0)] * world
        shares = [int(h["part"]["meta"][:, 1].sum()) if h["part"] is not None else 0 for h in heads]
    else:
        heads = None
    got = [None]
    dist.scatter_object_list(got, heads, src=0, group=group)
    head = got[0]
    part = head["part"]
    m = int(part["meta"][:, 1].sum()) if part is not None else 0
    if rank == 0:
        # the segments leave point to point; rank 0's own share is packed last (the others start while it is busy)
        pending = []
        for dst in range(1, world):
            if len(parts[dst]) == 0:
                continue
            bufs = _buffers(lib, shares[dst], dev)
            idx.frontier_pack(parts[dst], *bufs)
            for b in bufs:
                pending.append((dist.isend(b[:shares[dst]], dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group), b))
        if left > 0:      # (otherwise the run finished before it was wide enough to divide: nothing to hand out)
            bufs = _buffers(lib, m, dev)
            if m:
                idx.frontier_pack(parts[0], *bufs)
            idx.frontier_import(part, *bufs, minl=minl, minn=minn)
        res = idx.align_builtin_resume()
        for w, _ in pending:
            w.wait()
    else:
        res = empty_result(trace)
        if m:
            bufs = _buffers(lib, m, dev)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            for b in bufs:
                dist.recv(b[:m], src=src, group=group)
            if on_dev:
                # RCCL's recv only orders torch's current stream behind the transfer; the library copies from these buffers on
                # its own (non-blocking) stream, so the host has to see the transfers finished first
                torch.cuda.current_stream().synchronize()
            idx.frontier_import(part, *bufs, minl=minl, minn=minn, maxlcp=head["maxlcp"], trace=trace)
            res = idx.align_builtin_resume()
    out = [None] * world if rank == 0 else None
    dist.gather_object(res, out, dst=0, group=group)
    if rank != 0:
        return None
    merged = merge(out)
    merged["shares"] = shares
    return merged
