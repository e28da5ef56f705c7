"""One alignment over several GPUs: the frontier hand-off of include/reveal_amd.h.

SURVEY.md 8(e), second granularity.  The reference's recursion is a stack of
independent sub-indices popped by worker threads (reveallib/reveal.c:21-25,
:731-1338): children of a split cover disjoint text and disjoint ranges of the
shared inverse and are pushed after their parent's lower-casing (:1230-1234
before :1296).  Here the workers are GPUs.  Rank 0 builds the index and runs the
level loop until the frontier is wide enough; the frontier's sub-indices then
form a QUEUE of batches that the ranks PULL from (the reference's pop_index /
push_index with workers popping until the stack is empty, reveal.c:18-53,
interface.c:338-386): an idle rank asks rank 0 for the next batch and gets its
metadata plus the SA / LCP / BWT segments, 9 B per rank, point to point (RCCL
send/recv over xGMI with the nccl backend, host memory with gloo), finishes it
with the same level loop on a handle that only holds the text, and asks again;
rank 0 serves the requests between batches of its own, taken from the small end
of the queue.  Anchors are gathered on rank 0; their union is the anchor set of
the undivided run (tests/test_gpu_handoff.py compares it, the per-sub-index
trace and the lower-cased text with the oracle's).

No collective on the data path.  What stays on one GPU (construct + the first
levels) bounds the speed-up (Amdahl): at 2 x 250 Mbp the top of the recursion is
wide (every level streams all 5*10^8 ranks), so dividing pays; for small inputs
independent alignments per GPU (bench.py's default) are the better use of a node.
"""
import heapq

import time

import numpy as np

_CALLS = 0
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


def make_batches(sizes, world, per_rank=4):
    """the queue: sub-indices largest first, cut into batches of about total / (world * per_rank) ranks (a sub-index above that is a
    batch of its own: it is the unit of work, whoever takes it splits it further).  -> list of int32 arrays, largest batch first"""
    sizes = np.asarray(sizes, dtype=np.int64)
    if len(sizes) == 0:
        return []
    target = max(int(sizes.sum()) // max(world * per_rank, 1), 1)
    out, cur, load = [], [], 0
    for s in np.argsort(-sizes, kind="stable"):
        cur.append(int(s)); load += int(sizes[s])
        if load >= target:
            out.append(np.asarray(sorted(cur), dtype=np.int32)); cur, load = [], 0
    if cur:
        out.append(np.asarray(sorted(cur), dtype=np.int32))
    return out


def _queue_store(dist):
    """the key-value store behind the default process group (the requests of the work queue go through it)"""
    get = getattr(dist.distributed_c10d, "_get_default_store", None)
    if get is None:
        raise RuntimeError("reveal_amd.shard: this torch.distributed offers no access to the process group's store")
    return get()


def _send_batch(dist, group, dst, lib, head, meta_bytes, bufs, m, dev):
    import torch
    gdst = dist.get_global_rank(group, dst) if group is not None else dst
    dist.send(torch.tensor(head, dtype=torch.int64, device=dev), dst=gdst, group=group)
    if head[0] < 0:
        return
    dist.send(torch.frombuffer(bytearray(meta_bytes), dtype=torch.uint8).to(dev), dst=gdst, group=group)
    if m:
        for b in bufs:
            dist.send(b, dst=gdst, group=group)


def _elem_sizes(lib):
    return (8 if lib.sa64 else 4, 4, 1)


def align_sharded_group(idx, grp, mem, minl=20, minn=2, stop_subs=None, trace=False, per_rank=4):
    """align_sharded over reveal_amd.transport: `grp` = transport.Group (local sockets: requests, metadata, results), `mem` = where the segments live --
    transport.DeviceMemory (each rank its GPU: the owner exports its staging buffers once, workers copy their batches out device to device),
    transport.SharedMemory (the same code path without a device) or transport.HostMemory (segments inside the messages).  No torch, no collective.
    -> on rank 0 the merged result (as align_sharded), None on the others."""
    rank, world = grp.rank, grp.world
    lib = idx._lib
    sizes = _elem_sizes(lib)
    if world == 1:
        idx.construct()
        res = idx.align_builtin(minl, minn, trace=trace)
        res["shares"] = [int(idx.n)]; res["batches"] = [1]
        return res
    shared = mem.kind == "device"
    results = []
    if rank == 0:
        idx.construct()
        left, fr, _ = balanced_frontier(idx, world, stop_subs or 4 * world, minl, minn, trace=trace, tolerance=1.5)
        maxlcp = idx.maxlcp
        batches, staged, tokens, seeds = [], None, None, None
        if left > 0:
            batches = make_batches(fr["meta"][:, 1], world, per_rank)
            order = np.concatenate(batches)
            total = int(fr["meta"][order, 1].sum())
            staged = tuple(mem.alloc(total, sz) for sz in sizes)
            idx.frontier_pack(order, *staged)          # (synchronous: the copies have finished when it returns)
            seeds = [idx.frontier_seeds(b) for b in batches] if idx.picker_info()["kind"] == 1 else None
            if shared:
                tokens = tuple(mem.export(b) for b in staged)      # once: a worker opens them with its first batch
        first = [0]
        for b in batches:
            first.append(first[-1] + int(fr["meta"][b, 1].sum()))
        lo, hi = 0, len(batches)                       # the queue: workers take from the large end (lo), rank 0 from the small one
        shares, counts = [0] * world, [0] * world
        told = set()                                   # workers that have the tokens
        active = set(range(1, world))                  # workers that have not been told "no more" yet
        remote = {}

        def part_of(k):
            part = subset(fr, batches[k])
            if seeds is not None:
                part["seeds"] = seeds[k]
            return part

        def serve(w, msg):
            nonlocal lo
            if msg[0] == "result":
                remote[w] = msg[1]
                return
            assert msg[0] == "next", msg
            if lo >= hi:
                grp.send(w, None)
                active.discard(w)
                return
            k = lo; lo += 1
            m = first[k + 1] - first[k]
            reply = dict(part=part_of(k), m=m, first=first[k], maxlcp=maxlcp)
            if shared:
                if w not in told:
                    reply["tokens"] = tokens; told.add(w)
            else:
                reply["segments"] = tuple(mem.bytes_of(b[first[k]:first[k + 1]]) for b in staged)
            grp.send(w, reply)
            shares[w] += m; counts[w] += 1

        while active or lo < hi or len(remote) < world - 1:
            # requests first (a blocking wait when rank 0 has nothing of its own to do: no polling of a store, no sleep)
            busy = lo < hi
            got = grp.ready(0 if busy else None)
            for w, msg in got:
                serve(w, msg)
            if got or lo >= hi:
                continue
            hi -= 1
            k = hi
            bufs = tuple(b[first[k]:first[k + 1]] for b in staged)
            idx.frontier_import(part_of(k), *bufs, minl=minl, minn=minn, maxlcp=maxlcp, trace=trace)
            results.append(idx.align_builtin_resume())
            shares[0] += first[k + 1] - first[k]; counts[0] += 1
        if not results:
            if left > 0:
                empty = tuple(mem.alloc(0, sz) for sz in sizes)
                idx.frontier_import(subset(fr, np.zeros(0, np.int32)), *empty, minl=minl, minn=minn, maxlcp=maxlcp, trace=trace)
            results.append(idx.align_builtin_resume())
        mine = merge(results) if len(results) > 1 else results[0]
        merged = merge([mine] + [remote[w] for w in sorted(remote)])
        merged["shares"] = shares; merged["batches"] = counts
        for w in range(1, world):
            grp.send(w, "done")                        # (the staging buffers may go: every worker has copied what it took)
        mem.release()
        return merged
    opened = None
    while True:
        reply = grp.ask(("next",))
        if reply is None:
            break
        m = reply["m"]
        mine = tuple(mem.alloc(m, sz) for sz in sizes)
        if shared:
            if "tokens" in reply:
                opened = tuple(mem.open(t) for t in reply["tokens"])
            for dst, src in zip(mine, opened):
                mem.copy(dst, src[reply["first"]:reply["first"] + m], m)
        else:
            for dst, data in zip(mine, reply["segments"]):
                mem.fill(dst, data)
        idx.frontier_import(reply["part"], *mine, minl=minl, minn=minn, maxlcp=reply["maxlcp"], trace=trace)
        results.append(idx.align_builtin_resume())
    if not results:
        results.append(empty_result(trace))
    grp.send(0, ("result", merge(results) if len(results) > 1 else results[0]))
    assert grp.recv() == "done"
    mem.release()
    return None


def align_sharded(idx, minl=20, minn=2, stop_subs=None, group=None, trace=False, per_rank=4):
    """Divide ONE alignment over the ranks of `group` as a work queue.  Every rank passes an index that holds the same samples
    (addsample / addsequence done, construct not needed); rank 0's is constructed here.
    -> on rank 0 the merged result (shape of index.align_builtin, plus 'shares' = ranks each rank ended up finishing and
       'batches' = how many batches it took); None on the other ranks.
    (This form rides on torch.distributed -- RCCL send / recv with the nccl backend, host memory with gloo -- for callers that live in a
    process group already; align_sharded_group does the same over reveal_amd.transport with HIP's own inter-process copies and no torch.)"""
    import pickle
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
        res["batches"] = [1]
        return res

    def sync_dev():
        # RCCL's recv / send only order torch's current stream; the library copies on its own (non-blocking) stream
        if on_dev:
            torch.cuda.current_stream().synchronize()

    results, taken, nb = [], 0, 0
    global _CALLS
    _CALLS += 1                                        # (every rank calls in the same order: the same number everywhere)
    store = _queue_store(dist)
    # the queue's keys carry the owner's GLOBAL rank: two groups running at once never share a prefix, and a group that does not
    # contain global rank 0 works like any other
    owner = dist.get_global_rank(group, 0) if group is not None else 0
    key = "reveal_amd/queue/%d/%d/%%d/%%d" % (owner, _CALLS)
    if rank == 0:
        idx.construct()
        left, fr, _ = balanced_frontier(idx, world, stop_subs or 4 * world, minl, minn, trace=trace, tolerance=1.5)
        maxlcp = idx.maxlcp
        batches, staged = [], None
        if left > 0:
            batches = make_batches(fr["meta"][:, 1], world, per_rank)
            # every segment leaves the level arrays now, in queue order: the owner's own batches replace its frontier later
            order = np.concatenate(batches)
            total = int(fr["meta"][order, 1].sum())
            staged = _buffers(lib, total, dev)
            idx.frontier_pack(order, *staged)          # (synchronous: the copies have finished when it returns)
            # rv_set_picker(1): the seed lists of the sub-indices leave with them (the owner's frontier is replaced by its own batches later)
            seeds = [idx.frontier_seeds(b) for b in batches] if idx.picker_info()["kind"] == 1 else None
        first = [0]
        for b in batches:
            first.append(first[-1] + int(fr["meta"][b, 1].sum()))
        # requests: a key per worker and request in the process group's store (gloo's irecv does not report completion to a poll,
        # and a blocking receive would keep rank 0 from its own batches); the data itself travels point to point
        reqs = {w: 0 for w in range(1, world)}          # worker -> number of its next request
        lo, hi = 0, len(batches)                       # the queue: batches[lo:hi]; workers take from the large end (lo), rank 0 from the small one
        shares = [0] * world
        counts = [0] * world

        def hand_out(w, k):
            part = subset(fr, batches[k])
            if seeds is not None:
                part["seeds"] = seeds[k]
            m = first[k + 1] - first[k]
            bufs = tuple(b[first[k]:first[k + 1]] for b in staged)
            meta = pickle.dumps(part, protocol=4)
            _send_batch(dist, group, w, lib, [len(batches[k]), m, len(meta), maxlcp], meta, bufs, m, dev)
            shares[w] += m; counts[w] += 1

        idle = 0
        while reqs or lo < hi:
            served = False
            for w in list(reqs):
                if store.check([key % (w, reqs[w])]):
                    served = True
                    try:
                        store.delete_key(key % (w, reqs[w]))      # (a served request leaves nothing behind in the store)
                    except Exception:
                        pass
                    reqs[w] += 1
                    if lo < hi:
                        hand_out(w, lo); lo += 1
                    else:
                        _send_batch(dist, group, w, lib, [-1, 0, 0, 0], b"", (), 0, dev)
                        del reqs[w]
            if lo < hi and not served:                 # nobody is waiting: a batch for rank 0 itself, from the small end
                hi -= 1
                k = hi
                part = subset(fr, batches[k])
                if seeds is not None:
                    part["seeds"] = seeds[k]
                bufs = tuple(b[first[k]:first[k + 1]] for b in staged)
                idx.frontier_import(part, *bufs, minl=minl, minn=minn, maxlcp=maxlcp, trace=trace)
                results.append(idx.align_builtin_resume())
                shares[0] += first[k + 1] - first[k]; counts[0] += 1
                idle = 0
            elif served:
                idle = 0
            else:
                # the queue is empty and the workers are busy with their last batches: wait for their final requests without hammering
                # the store (50 us at first, 2 ms after a while)
                idle += 1
                time.sleep(min(0.002, 0.00005 * idle))
        if not results:
            # rank 0 took no batch (the run finished before it was wide enough to divide, or the workers emptied the queue): its
            # anchors of the levels in front of the hand-off are collected by finishing an empty frontier
            if left > 0:
                idx.frontier_import(subset(fr, np.zeros(0, np.int32)), *_buffers(lib, 0, dev), minl=minl, minn=minn, maxlcp=maxlcp, trace=trace)
            results.append(idx.align_builtin_resume())
    else:
        src = dist.get_global_rank(group, 0) if group is not None else 0
        while True:
            store.set(key % (rank, nb), "1")
            head = torch.zeros(4, dtype=torch.int64, device=dev)
            dist.recv(head, src=src, group=group)
            nsubs, m, nmeta, maxlcp = (int(x) for x in head.tolist())
            if nsubs < 0:
                break
            mt = torch.empty(nmeta, dtype=torch.uint8, device=dev)
            dist.recv(mt, src=src, group=group)
            part = pickle.loads(mt.cpu().numpy().tobytes())
            bufs = _buffers(lib, m, dev)
            if m:
                for b in bufs:
                    dist.recv(b[:m], src=src, group=group)
            sync_dev()
            idx.frontier_import(part, *bufs, minl=minl, minn=minn, maxlcp=maxlcp, trace=trace)
            results.append(idx.align_builtin_resume())
            taken += m; nb += 1
        if not results:
            results.append(empty_result(trace))
    # the owner's anchors of the levels in front of the hand-off ride in its first resume(): a handle with a run in progress keeps them
    mine = merge(results) if len(results) > 1 else results[0]
    out = [None] * world if rank == 0 else None
    dist.gather_object(mine, out, dst=owner, group=group)      # (dst is a global rank)
    if rank != 0:
        return None
    merged = merge(out)
    merged["shares"] = shares
    merged["batches"] = counts
    return merged
