"""One alignment divided over several index handles (include/reveal_amd.h "frontier hand-off",
reveal_amd/shard.py; SURVEY.md 8(e) second granularity): the owner stops once its frontier is wide
enough, shares of the sub-indices go to handles that only hold the text, every handle finishes its
share.  The union must be what the undivided recursion of the oracle visits: same sub-indices (sizes,
intervals, SA / LCP hashes, scan results, picks), same anchors, same lower-cased text."""
import numpy as np
import pytest

from helpers import assemble, fa, feed, synth
from reveal_amd import shard
from test_gpu_align import FIELDS, mod, oracle_run

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def level_pipeline_only(monkeypatch, request):
    """the hand-off divides the level pipeline's frontier; an untraced two-sample run would be finished by the anchor cascade
    before there is one (rv_align_builtin_until returns 0 then: test_cascade_leaves_nothing_to_divide)"""
    if "cascade" not in request.node.name:
        monkeypatch.setenv("RV_NO_CASCADE", "1")


def divided(inputs, minl, minn, stop_subs, nparts, sa64=False, trace=True, device_buffers=False, picker=None):
    M = mod(sa64)
    owner = feed(M.index(), inputs)
    if picker is not None:
        owner.set_picker(picker)
    owner.construct()
    lib = owner._lib
    left = owner.align_builtin_until(stop_subs, minl, minn, trace=trace)
    results, shares = [], []
    if left > 0:
        fr = owner.frontier()
        assert len(fr["meta"]) == left >= stop_subs
        parts = shard.partition(fr["meta"][:, 1], nparts)
        packed = []
        for subs in parts:
            part = shard.subset(fr, subs)
            if picker is not None:
                part["seeds"] = owner.frontier_seeds(subs)
            m = int(part["meta"][:, 1].sum())
            if device_buffers:
                bufs = shard._buffers(lib, m, "cuda:0")
            else:
                bufs = (np.zeros(max(m, 1), lib.sa_t), np.zeros(max(m, 1), lib.lcp_t), np.zeros(max(m, 1), np.uint8))
            assert owner.frontier_pack(subs, *bufs) == m
            packed.append((part, bufs))
            shares.append(m)
        assert sum(shares) == fr["m"]
        owner.frontier_import(packed[0][0], *packed[0][1], minl=minl, minn=minn)
        for part, bufs in packed[1:]:
            w = feed(M.index(), inputs)          # a worker: samples only, no construct
            if picker is not None:
                w.set_picker(picker)
            if len(part["meta"]) == 0:
                results.append(shard.empty_result(trace))
                continue
            w.frontier_import(part, *bufs, minl=minl, minn=minn, maxlcp=owner.maxlcp, trace=trace)
            results.append(w.align_builtin_resume())
    results.insert(0, owner.align_builtin_resume())
    return shard.merge(results), shares, left


def check(inputs, minl=20, minn=2, stop_subs=4, nparts=3, sa64=False, device_buffers=False):
    ref, T = oracle_run(inputs, minl, minn, sa64)
    got, shares, left = divided(inputs, minl, minn, stop_subs, nparts, sa64, True, device_buffers)
    rt, gt = ref["trace"], got["trace"]
    assert len(rt) == len(gt), (len(rt), len(gt))
    ro = np.lexsort((rt["key"], rt["depth"])); go = np.lexsort((gt["key"], gt["depth"]))
    for f in FIELDS:
        a, b = rt[f][ro].astype(np.uint64), gt[f][go].astype(np.uint64)
        assert np.array_equal(a, b), "field %s differs at %d of %d sub-indices" % (f, int((a != b).sum()), len(a))
    rl, rn, roff, rpos = ref["anchors"]
    ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    gl, goff, gpos = got["anchors"]
    ga = sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl)))
    assert ra == ga
    assert shard.lower_text(T, got["anchors"]).tobytes() == ref["T"]
    assert got["stats"]["splits"] == ref["stats"]["nsplits"]
    assert got["stats"]["steps"] == ref["stats"]["nsteps"]
    return shares, left


@pytest.mark.parametrize("name,inputs,minl,stop,parts", [
    ("1a1b", fa("1a", "1b"), 20, 4, 3),
    ("1a1b_wide", fa("1a", "1b"), 10, 16, 4),
    ("1e1b", fa("1e", "1b"), 20, 2, 2),
    ("d1d2_never_wide", fa("d1", "d2"), 20, 1000, 2),      # the run ends before the frontier is that wide
])
def test_divided_pair(name, inputs, minl, stop, parts):
    check(inputs, minl, 2, stop, parts)


def test_divided_pair_64bit():
    check(fa("1a", "1b"), 20, 2, 4, 2, sa64=True)


@pytest.mark.parametrize("L,stop,parts", [(200000, 8, 8), (1000000, 32, 4)])
def test_divided_synthetic_pair(L, stop, parts):
    seqs = [g.decode() for g in synth.genomes(L, 2)]
    shares, left = check(seqs, 20, 2, stop, parts)
    assert left >= stop and len(shares) == parts and max(shares) < 2 * (sum(shares) // parts + max(shares) // 4 + 1)


def test_divided_multi():
    check(fa("1a", "1b", "1c"), 20, 3, 4, 3)
    seqs = [g.decode() for g in synth.genomes(60000, 5)]
    check(seqs, 20, 5, 8, 4)


def test_divided_device_buffers():
    """segments packed into / imported from device memory (what the nccl transport uses)"""
    pytest.importorskip("torch")
    seqs = [g.decode() for g in synth.genomes(300000, 2)]
    check(seqs, 20, 2, 8, 3, device_buffers=True)


def test_divided_untraced_matches_undivided():
    """the production configuration (no trace: device-side picker, decisions, early split, leaf kernel)"""
    seqs = [g.decode() for g in synth.genomes(1500000, 2)]
    M = mod(False)
    idx = feed(M.index(), seqs)
    idx.construct()
    one = idx.align_builtin(20, 2)
    got, shares, left = divided(seqs, 20, 2, 16, 4, trace=False)
    assert left >= 16

    def aset(r):
        l, off, pos = r["anchors"]
        return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))
    assert aset(one) == aset(got)
    for k in ("steps", "splits", "anchored_bp"):
        assert one["stats"][k] == got["stats"][k], k
    assert shard.lower_text(assemble(seqs)[0], got["anchors"]).tobytes() == idx.T.encode("latin-1")


@pytest.mark.parametrize("genomes,L,seedsize", [(2, 300000, 40), (3, 120000, 40), (2, 200000, 10000)])
def test_divided_with_the_native_picker(genomes, L, seedsize):
    """rv_set_picker(1) through the hand-off: the workers' scans hand whole lists to the chain picker, and the seed lists the owner's picker
    calls left for the frontier's sub-indices (the reference's skipmums, reveal.c:1157, 1180) travel with them -- same anchors, in the same
    number of picker calls and seeded calls, as the undivided run"""
    from reveal_amd import schemes
    seqs = [g.decode() for g in synth.genomes(L, genomes, indelfrac=0.2)]
    args = schemes.PickerArgs(seedsize=seedsize)
    M = mod(False)
    idx = feed(M.index(), seqs)
    idx.set_picker(args)
    idx.construct()
    one = idx.align_builtin(20, 2)
    info = idx.picker_info()
    assert info["kind"] == 1 and (seedsize > 1000 or info["seeded"] > 0)
    got, shares, left = divided(seqs, 20, 2, 8, 3, trace=False, picker=args)
    assert left >= 8

    def aset(r):
        l, off, pos = r["anchors"]
        return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))
    assert aset(one) == aset(got)
    for k in ("steps", "splits", "anchored_bp"):
        assert one["stats"][k] == got["stats"][k], k


TWO_RANK = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch, torch.distributed as dist
from reveal_amd import _lib, reveallib, shard, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")              # both ranks share the one GPU of the test box; segments travel through host memory
_lib.set_device(0)
seqs = synth.genomes(%d, %d, seed=7)

def feed(idx):
    for g in seqs:
        idx.addsample("s"); idx.addsequence(g.decode())
    return idx

res = None
for it in range(2):                           # handles are reused from step to step, as bench.py --divide does
    idx = feed(reveallib.index()) if it == 0 else idx
    res = shard.align_sharded(idx, 20, 2, stop_subs=8)
if rank == 0:
    os.environ["RV_NO_CASCADE"] = "1"        # (the undivided reference through the same level pipeline the shares run)
    one = feed(reveallib.index()); one.construct()
    ref = one.align_builtin(20, 2)
    def aset(r):
        l, off, pos = r["anchors"]
        return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))
    T0 = np.frombuffer(b"$".join(seqs) + b"$", dtype=np.uint8)
    print(json.dumps({"same_anchors": aset(res) == aset(ref), "anchors": len(ref["anchors"][0]), "shares": res["shares"], "batches": res["batches"],
                      "same_text": shard.lower_text(T0.tobytes(), res["anchors"]).tobytes() == one.T.encode("latin-1"),
                      "same_counts": all(res["stats"][k] == ref["stats"][k] for k in ("steps", "splits", "anchored_bp"))}))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,genomes,L", [(2, 2, 400000), (3, 2, 600000), (3, 4, 150000)])
def test_processes_divide_one_alignment(tmp_path, world, genomes, L):
    """shard.align_sharded end to end: two and three processes (gloo; they share the one GPU of the test box), rank 0 constructs,
    runs the top levels and serves the queue of sub-index batches the ranks pull from -- the workers only hold the text; the
    merged result is the undivided one.  Three ranks, uneven sub-indices, two and four samples."""
    import json, os, subprocess, sys
    from helpers import ROOT
    script = tmp_path / "ranks.py"
    script.write_text(TWO_RANK % (ROOT, L, genomes))
    port = str(29530 + world + genomes)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert r["same_anchors"] and r["same_text"] and r["same_counts"] and r["anchors"] > 500
    assert len(r["shares"]) == world and sum(1 for x in r["shares"] if x > 0) >= 2 and sum(r["batches"]) >= world


IPC_RANK = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from reveal_amd import _lib, reveallib, shard, synth, transport
assert "torch" not in sys.modules                 # the hand-off over reveal_amd.transport needs no tensor library
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
_lib.set_device(0)                                # (the ranks share the one GPU of the test box: an IPC handle opens in another process all the same)
seqs = synth.genomes(%d, %d, seed=7)
grp = transport.Group.from_env()

def feed(idx):
    for g in seqs:
        idx.addsample("s"); idx.addsequence(g.decode())
    return idx

res = None
for it in range(2):
    idx = feed(reveallib.index()) if it == 0 else idx
    res = shard.align_sharded_group(idx, grp, transport.DeviceMemory(idx._lib, 0), 20, 2, stop_subs=8)
if rank == 0:
    os.environ["RV_NO_CASCADE"] = "1"
    one = feed(reveallib.index()); one.construct()
    ref = one.align_builtin(20, 2)
    def aset(r):
        l, off, pos = r["anchors"]
        return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))
    T0 = np.frombuffer(b"$".join(seqs) + b"$", dtype=np.uint8)
    print(json.dumps({"same_anchors": aset(res) == aset(ref), "anchors": len(ref["anchors"][0]), "shares": res["shares"], "batches": res["batches"],
                      "same_text": shard.lower_text(T0.tobytes(), res["anchors"]).tobytes() == one.T.encode("latin-1"),
                      "same_counts": all(res["stats"][k] == ref["stats"][k] for k in ("steps", "splits", "anchored_bp"))}))
grp.barrier(); grp.close()
'''


@pytest.mark.parametrize("world,genomes,L", [(2, 2, 400000), (3, 4, 150000)])
def test_processes_divide_one_alignment_over_hip_ipc(tmp_path, world, genomes, L):
    """shard.align_sharded_group with transport.DeviceMemory: no torch in the ranks -- requests over local sockets, the owner's staging buffers exported
    once with hipIpcGetMemHandle, every worker process opens them and copies its batches out device to device (point-to-point copies of
    child arrays, no collective).  The merged result is the undivided one."""
    import json, os, subprocess, sys
    from helpers import ROOT
    script = tmp_path / "ranks_ipc.py"
    script.write_text(IPC_RANK % (ROOT, L, genomes))
    port = str(29560 + world + genomes)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    r = json.loads([x for x in outs[0][0].splitlines() if x.startswith("{")][-1])
    assert r["same_anchors"] and r["same_text"] and r["same_counts"] and r["anchors"] > 500, r
    assert len(r["shares"]) == world and sum(1 for x in r["shares"] if x > 0) >= 2 and sum(r["batches"]) >= world, r


def test_processes_divide_one_alignment_with_the_cascade_on(tmp_path):
    """the default path (no RV_NO_CASCADE in the ranks' environment; this test's name keeps the fixture away): four samples on three ranks --
    rv_align_builtin_until runs the interval cascade first, what it leaves undecided becomes the frontier (install_frontier), balanced_frontier
    widens it, the queue hands it out; the cascade's own anchors ride in rank 0's first resume().  Anchors and text equal the undivided run's
    (its count of visited sub-indices is the level pipeline's, which also visits children the cascade decides without visiting: not compared)"""
    import json, os, subprocess, sys
    from helpers import ROOT
    script = tmp_path / "ranks.py"
    script.write_text(TWO_RANK % (ROOT, 150000, 4))
    port = "29547"
    env = {k: v for k, v in os.environ.items() if not k.startswith("RV_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert r["same_anchors"] and r["same_text"] and r["anchors"] > 500
    assert len(r["shares"]) == 3


def test_widened_frontier_divides_evenly():
    """shard.balanced_frontier: the owner goes on level by level (rv_align_builtin_continue) until no share of the
    largest-first partition exceeds the mean by more than the tolerance; the divided result is still the undivided one"""
    seqs = [g.decode() for g in synth.genomes(1500000, 2, seed=5)]
    M = mod(False)
    one = feed(M.index(), seqs)
    one.construct()
    ref = one.align_builtin(20, 2)
    owner = feed(M.index(), seqs)
    owner.construct()
    world = 4
    left, fr, parts = shard.balanced_frontier(owner, world, 2, 20, 2, tolerance=1.10)
    assert left > 2 and len(parts) == world
    sizes = fr["meta"][:, 1]
    loads = np.array([int(sizes[p].sum()) for p in parts], dtype=np.float64)
    assert loads.max() <= 1.10 * loads.mean() or left >= 64 * world
    lib = owner._lib
    results = []
    packed = []
    for subs in parts:
        part = shard.subset(fr, subs)
        m = int(part["meta"][:, 1].sum())
        bufs = (np.zeros(max(m, 1), lib.sa_t), np.zeros(max(m, 1), lib.lcp_t), np.zeros(max(m, 1), np.uint8))
        assert owner.frontier_pack(subs, *bufs) == m
        packed.append((part, bufs))
    owner.frontier_import(packed[0][0], *packed[0][1], minl=20, minn=2)
    for part, bufs in packed[1:]:
        w = feed(M.index(), seqs)
        w.frontier_import(part, *bufs, minl=20, minn=2, maxlcp=owner.maxlcp)
        results.append(w.align_builtin_resume())
    results.insert(0, owner.align_builtin_resume())
    got = shard.merge(results)

    def aset(r):
        l, off, pos = r["anchors"]
        return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))
    assert aset(got) == aset(ref)
    for k in ("steps", "splits", "anchored_bp"):
        assert got["stats"][k] == ref["stats"][k], k
    # continue on a finished / never-started run is a no-op
    assert one.align_builtin_continue(8) == 0


def test_cascade_leaves_nothing_to_divide():
    """an untraced two-sample run the anchor cascade decides: align_builtin_until finishes it (frontier size 0), the result is the
    undivided one -- what shard.align_sharded then does is gather that result"""
    seqs = [g.decode() for g in synth.genomes(300000, 2, seed=5)]
    M = mod(False)
    one = feed(M.index(), seqs)
    one.construct()
    ref = one.align_builtin(20, 2)
    idx = feed(M.index(), seqs)
    idx.construct()
    assert idx.align_builtin_until(4, 20, 2) == 0 and idx.cascade_info()["done"]
    got = idx.align_builtin_resume()

    def aset(r):
        l, off, pos = r["anchors"]
        return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))
    assert aset(got) == aset(ref) and got["stats"]["steps"] == ref["stats"]["steps"]


def _aset(r):
    l, off, pos = r["anchors"]
    return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))


def test_a_held_result_survives_a_later_stopped_run():
    """A result the caller still holds lives in arrays the library had page-locked and written directly (rv_set_result_buffers).  A later
    run on the same handle through align_builtin_until / _resume (other minl: other anchors) must neither write into those arrays nor hand
    out stale staging data: the held arrays stay what they were, the new result is the oracle's."""
    seqs = [g.decode() for g in synth.genomes(500000, 2, seed=5)]
    M = mod(False)
    idx = feed(M.index(), seqs)
    idx.construct()
    r1 = idx.align_builtin(20, 2)
    want20 = _aset(r1)
    del r1                                         # the arrays go back to the index ...
    idx.construct()
    r2 = idx.align_builtin(20, 2)                  # ... and this run is delivered straight into them
    assert _aset(r2) == want20
    snap = tuple(np.array(x, copy=True) for x in r2["anchors"])
    idx.construct()
    left = idx.align_builtin_until(8, 31, 2)
    assert left >= 8
    r3 = idx.align_builtin_resume()
    for a, b in zip(r2["anchors"], snap):
        assert np.array_equal(a, b), "a run overwrote the arrays of a result the caller still holds"
    ref, _ = oracle_run(seqs, 31, 2, False)
    rl, rn, roff, rpos = ref["anchors"]
    assert _aset(r3) == sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
    assert _aset(r3) != want20


def test_result_buffers_cleared_between_the_run_and_its_fetch():
    """C-ABI contract (include/reveal_amd.h at rv_set_result_buffers): clearing the arrays after a run has delivered into them and before
    rv_fetch_anchors keeps the result -- it moves to the library's staging buffer."""
    import ctypes
    from reveal_amd import _lib
    from reveal_amd._index import _page_array
    seqs = [g.decode() for g in synth.genomes(300000, 2, seed=6)]
    M = mod(False)
    idx = feed(M.index(), seqs)
    idx.construct()
    want = _aset(idx.align_builtin(20, 2))
    dll, h = idx._dll, idx._h
    na = len(want)
    l = _page_array(na + 8, np.uint32); off = _page_array(na + 9, np.int64); pos = _page_array(2 * na + 16, np.int64)
    idx.construct()
    assert dll.rv_set_result_buffers(h, l.ctypes.data, len(l), off.ctypes.data, len(off), pos.ctypes.data, len(pos)) == 0
    st = _lib.RvAlignStats()
    assert dll.rv_align_builtin(h, 20, 2, ctypes.byref(st)) == 0
    assert dll.rv_set_result_buffers(h, None, 0, None, 0, None, 0) == 0          # cleared before the fetch
    l[:] = 0; off[:] = 0; pos[:] = 0
    l2 = np.zeros(na, np.uint32); off2 = np.zeros(na + 1, np.int64); pos2 = np.zeros(2 * na, np.int64)
    mem = ctypes.c_int64(0)
    assert dll.rv_anchor_count(h, ctypes.byref(mem)) == na and mem.value == 2 * na
    assert dll.rv_fetch_anchors(h, l2.ctypes.data, off2.ctypes.data, pos2.ctypes.data) == 0
    assert _aset(dict(anchors=(l2, off2, pos2))) == want
