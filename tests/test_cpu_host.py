"""CPU suite: host-side logic of the package, the C-ABI surface, and the multi-process
bench plumbing (gloo, world_size 2).  No compute call reaches a GPU here."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, assemble, fa, synth
from reveal_amd import rem, shard


def test_abi_exports_every_declared_symbol():
    """both libraries load and export exactly what include/reveal_amd.h declares"""
    from reveal_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "reveal_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for sa64 in (False, True):
        L = _lib.get(sa64)                      # resolves every symbol with its argtypes
        assert L.dll.rv_sa_bits() == (64 if sa64 else 32)
        assert L.dll.rv_abi_version() >= 1
        for sym in declared:
            assert hasattr(L.dll, sym)


def test_no_gpu_means_loud_failure():
    """there is no CPU fallback: without a device index() raises the module's error type"""
    from reveal_amd import _lib, reveallib
    if _lib.get(False).dll.rv_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(reveallib.error, match="no HIP device"):
        reveallib.index()


def test_product_never_touches_the_oracle():
    """nothing under reveal_amd/ imports, loads or links oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "reveal_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "oracle_ctypes" not in txt and "ref_ctypes" not in txt, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), os.path.join(dirpath, f)


def test_fasta_protocol():
    """reveal/utils.py:304-375: one sample per file, one '$'-terminated sequence per contig, upper-cased"""
    T, nsep, nodes = assemble(fa("1e", "1b"))
    assert T.count(b"$") == 4 and len(nodes) == 4 and len(nsep) == 1
    assert nodes[0][0] == 0 and all(T[e:e + 1] == b"$" for _, e in nodes)
    assert nsep[0] == nodes[2][1]                       # the '$' closing the last contig of the first file
    assert T == T.upper()
    names = [n for n, _ in rem.fasta_reader(fa("t1")[0])]
    assert names == ["t1.1", "t1.2"]


def test_synthetic_generator_is_deterministic_and_exact():
    a = synth.genomes(100000, 3, seed=42)
    b = synth.genomes(100000, 3, seed=42)
    assert a == b and len(a) == 3 and all(len(x) == 100000 for x in a)
    x0 = np.frombuffer(a[0], dtype=np.uint8)
    for k in (1, 2):
        xk = np.frombuffer(a[k], dtype=np.uint8)
        assert int((x0 != xk).sum()) == 1000            # exactly floor(L/100) substitutions, all to a different base
    assert set(a[0]) <= set(b"ACGT")
    assert synth.genomes(1000, 2, seed=43)[0] != synth.genomes(1000, 2, seed=42)[0]


class FakeIdx:
    def __init__(self, nodes, nsamples):
        self.nodes, self.nsamples = set(nodes), nsamples


def test_bench_callbacks_contract():
    """mumpicker -> () | (mum, skipleft, skipright); graphalign -> 7-tuple (reveal.c:839-999)"""
    idx = FakeIdx([(0, 100), (101, 200)], 2)
    mums = [(20, 2, ((0, 10), (1, 120))), (30, 2, ((0, 50), (1, 150))), (30, 2, ((0, 5), (1, 160))), (40, 1, ((0, 70),))]
    mum, sl, sr = rem.bench_mumpicker(mums, idx, precomputed=False, minlength=20)
    assert mum == (30, 2, ((0, 5), (1, 160))) and sl == [] and sr == []       # longest full match, tie -> smallest coordinate
    assert rem.bench_mumpicker([(40, 1, ((0, 70),))], idx) == ()
    lead, trail, match, rest, merged, nl, nr = rem.linear_graphalign(idx, mum)
    assert lead == [(0, 5), (101, 160)] and trail == [(35, 100), (190, 200)] and match == [(5, 35), (160, 190)] and rest == []
    assert rem.linear_graphalign(idx, (30, 2, ((0, 90), (1, 150)))) is None    # would cross an interval end


def test_shard_partition_subset_merge_lower():
    from reveal_amd import shard
    sizes = [5, 100, 7, 40, 40, 1, 60]
    parts = shard.partition(sizes, 3)
    assert sorted(int(x) for p in parts for x in p) == list(range(len(sizes)))       # every sub-index exactly once
    loads = [sum(sizes[int(s)] for s in p) for p in parts]
    assert max(loads) == 100 and min(loads) >= 73                                     # largest first into the lightest share
    assert all(list(p) == sorted(p) for p in parts)
    assert [len(p) for p in shard.partition([3, 2], 4)] == [1, 1, 0, 0]                # fewer sub-indices than ranks
    fr = dict(level=3, m=sum(sizes), meta=np.array([[0, n, 2, 2, 0, -1] for n in sizes], dtype=np.int64),
              node_first=np.array([0, 2, 4, 5, 7, 9, 10, 12], dtype=np.int64), nodes=np.arange(24, dtype=np.int64).reshape(12, 2))
    sub = shard.subset(fr, [1, 2, 6])
    assert sub["meta"][:, 1].tolist() == [100, 7, 60] and sub["node_first"].tolist() == [0, 2, 3, 5]
    assert sub["nodes"][:, 0].tolist() == [4, 6, 8, 20, 22] and sub["level"] == 3
    a = dict(stats=dict(steps=3, splits=2, anchored_bp=30, scanned_ranks=10, levels=4, maxdepth=3, t_scan=.1, t_host=.1, t_split=.1, t_bubble=.1),
             anchors=(np.array([10, 20], np.uint32), np.array([0, 2, 4]), np.array([1, 50, 20, 70])), trace=None)
    b = dict(stats=dict(steps=1, splits=1, anchored_bp=5, scanned_ranks=4, levels=6, maxdepth=5, t_scan=.2, t_host=0, t_split=0, t_bubble=0),
             anchors=(np.array([5], np.uint32), np.array([0, 2]), np.array([40, 95])), trace=None)
    m = shard.merge([a, shard.empty_result(), b])
    assert m["anchors"][0].tolist() == [10, 20, 5] and m["anchors"][1].tolist() == [0, 2, 4, 6] and m["anchors"][2].tolist() == [1, 50, 20, 70, 40, 95]
    assert m["stats"]["steps"] == 4 and m["stats"]["levels"] == 6 and m["stats"]["anchored_bp"] == 35
    T = b"A" * 45 + b"$" + b"C" * 54
    low = shard.lower_text(T, m["anchors"]).tobytes()
    assert low[:1] == b"A" and low[1:11] == b"a" * 10 and low[11:20] == b"A" * 9 and low[20:45] == b"a" * 25 and low[45:46] == b"$"
    assert low[46:50] == b"C" * 4 and low[50:60] == b"c" * 10 and low[60:70] == b"C" * 10 and low[70:90] == b"c" * 20 and low[95:] == b"c" * 5


SHARD_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from reveal_amd import shard
world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
GROUP = None
if os.environ.get("SHARD_SUBGROUP"):                  # a group without global rank 0: ranks 1..world-1
    GROUP = dist.new_group(list(range(1, world)))
    if dist.get_rank() == 0:
        dist.barrier(); dist.destroy_process_group(); sys.exit(0)
rank = dist.get_rank(GROUP)                            # rank inside the group that divides the alignment

class FakeLib: sa64 = False
class FakeIndex:
    """the methods shard.align_sharded drives, on numpy: a frontier of 7 sub-indices whose 'recursion' turns every sub-index
    into one anchor (l = its size, members = checksums of the segments it was handed)"""
    _lib = FakeLib()
    n = 1000
    maxlcp = 77
    def __init__(self): self.constructed = False; self.front = None; self.done = []
    def construct(self): self.constructed = True
    def picker_info(self): return {"kind": 0}
    def align_builtin_until(self, stop, minl, minn, trace=False):
        assert self.constructed and rank == 0
        self.sizes = np.array([5, 100, 7, 40, 40, 1, 60])
        self.SA = np.arange(self.sizes.sum(), dtype=np.int32) * 3 + 1
        self.done = [(9, (0, 500))]                    # an anchor of the levels in front of the hand-off
        return len(self.sizes)
    def frontier(self):
        off = np.concatenate([[0], np.cumsum(self.sizes)[:-1]])
        meta = np.stack([off, self.sizes, np.full(7, 2), np.full(7, 2), np.zeros(7, int), np.full(7, -1)], axis=1).astype(np.int64)
        return dict(level=3, m=int(self.sizes.sum()), meta=meta, node_first=np.arange(8, dtype=np.int64), nodes=np.stack([off, off + self.sizes], axis=1).astype(np.int64))
    def frontier_pack(self, subs, sa, lcp, bwt):
        at = 0
        fr = self.frontier()
        for s in subs:
            o, n = int(fr["meta"][s, 0]), int(fr["meta"][s, 1])
            sa[at:at + n] = torch.from_numpy(self.SA[o:o + n]); lcp[at:at + n] = 7; bwt[at:at + n] = 65
            at += n
        return at
    def frontier_import(self, part, sa, lcp, bwt, minl=20, minn=2, maxlcp=None, trace=False):
        assert rank == 0 or (maxlcp == 77 and not self.constructed)
        self.front = (part, sa.numpy().copy(), lcp.numpy().copy(), bwt.numpy().copy())
    def align_builtin_continue(self, stop):
        return len(self.sizes)
    def align_builtin_resume(self):
        part, sa, lcp, bwt = self.front
        at = 0
        import time; time.sleep(0.02 * len(part["meta"]))      # (a batch takes a while: the other ranks get to ask meanwhile)
        for k in range(len(part["meta"])):
            n = int(part["meta"][k, 1])
            assert (lcp[at:at + n] == 7).all() and (bwt[at:at + n] == 65).all()
            assert part["nodes"][part["node_first"][k], 1] - part["nodes"][part["node_first"][k], 0] == n
            self.done.append((n, (int(sa[at:at + n].sum()), int(part["nodes"][part["node_first"][k], 0]))))
            at += n
        l = np.array([d[0] for d in self.done], np.uint32); pos = np.array([p for d in self.done for p in d[1]], np.int64)
        st = shard.empty_result()["stats"]; st["splits"] = len(self.done)
        self.done = []                                  # (a later import starts a new run on this handle)
        return dict(stats=st, anchors=(l, np.arange(0, 2 * len(l) + 1, 2), pos), trace=None)

import torch
res = shard.align_sharded(FakeIndex(), 20, 2, stop_subs=4, per_rank=2, group=GROUP)
if rank == 0:
    l, off, pos = res["anchors"]
    print(json.dumps({"anchors": sorted((int(l[k]), [int(x) for x in pos[off[k]:off[k + 1]]]) for k in range(len(l))), "shares": res["shares"], "batches": res["batches"], "splits": res["stats"]["splits"]}))
else:
    assert res is None
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_divided_alignment_protocol_gloo(tmp_path, world):
    """shard.align_sharded on two and three ranks (gloo, host memory) around a stand-in index: the frontier becomes a queue of
    batches the ranks pull from -- every sub-index reaches exactly one rank with its own segments and metadata, uneven sizes
    end up on different ranks, rank 0 serves between batches of its own and ends up with all anchors (also the one made in
    front of the hand-off)"""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % ROOT)
    port = str(29519 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    sizes = [5, 100, 7, 40, 40, 1, 60]
    SA = np.arange(sum(sizes)) * 3 + 1
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    want = sorted([(9, [0, 500])] + [(n, [int(SA[o:o + n].sum()), int(o)]) for o, n in zip(off, sizes)])
    assert [tuple(a) for a in r["anchors"]] == [(l, p) for l, p in want]
    assert r["splits"] == 8 and sum(r["shares"]) == sum(sizes) and len(r["shares"]) == world
    # the queue: batches of about total / (world * 2) ranks, largest first
    nbatches = len(shard.make_batches(sizes, world, 2))
    assert sum(r["batches"]) == nbatches and nbatches >= world
    assert sum(1 for x in r["shares"] if x > 0) >= 2                      # more than one rank took part


def test_divided_alignment_in_a_group_without_global_rank_0(tmp_path):
    """align_sharded(group=...) on a sub-group made of global ranks 1 and 2 of a world of three: the queue's keys are named after
    the owner's global rank, the owner is the group's rank 0"""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % ROOT)
    port = "29531"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SHARD_SUBGROUP="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert r["splits"] == 8 and len(r["shares"]) == 2 and sum(r["shares"]) == 253 and all(x > 0 for x in r["shares"])


GROUP_WORKER = r'''
import os, sys, json, ctypes
sys.path.insert(0, %r)
import numpy as np
from reveal_amd import shard, transport
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
grp = transport.Group.from_env()
mem = transport.SharedMemory() if os.environ["SHARD_MEM"] == "shared" else transport.HostMemory()

def arr(view, dtype):      # the numpy face of a transport buffer
    return np.frombuffer((ctypes.c_uint8 * (view.n * view.itemsize)).from_address(view.ptr), dtype=dtype)

class FakeLib: sa64 = False
class FakeIndex:
    """the methods shard.align_sharded_group drives, on numpy: a frontier of 7 sub-indices whose 'recursion' turns every sub-index
    into one anchor (l = its size, members = checksums of the segments it was handed)"""
    _lib = FakeLib()
    n = 1000
    maxlcp = 77
    def __init__(self): self.constructed = False; self.front = None; self.done = []
    def construct(self): self.constructed = True
    def picker_info(self): return {"kind": 0}
    def align_builtin_until(self, stop, minl, minn, trace=False):
        assert self.constructed and rank == 0
        self.sizes = np.array([5, 100, 7, 40, 40, 1, 60])
        self.SA = np.arange(self.sizes.sum(), dtype=np.int32) * 3 + 1
        self.done = [(9, (0, 500))]                    # an anchor of the levels in front of the hand-off
        return len(self.sizes)
    def frontier(self):
        off = np.concatenate([[0], np.cumsum(self.sizes)[:-1]])
        meta = np.stack([off, self.sizes, np.full(7, 2), np.full(7, 2), np.zeros(7, int), np.full(7, -1)], axis=1).astype(np.int64)
        return dict(level=3, m=int(self.sizes.sum()), meta=meta, node_first=np.arange(8, dtype=np.int64), nodes=np.stack([off, off + self.sizes], axis=1).astype(np.int64))
    def frontier_pack(self, subs, sa, lcp, bwt):
        assert sa.itemsize == 4 and lcp.itemsize == 4 and bwt.itemsize == 1
        at = 0
        fr = self.frontier()
        for s in subs:
            o, n = int(fr["meta"][s, 0]), int(fr["meta"][s, 1])
            arr(sa, np.int32)[at:at + n] = self.SA[o:o + n]; arr(lcp, np.int32)[at:at + n] = 7; arr(bwt, np.uint8)[at:at + n] = 65
            at += n
        return at
    def frontier_import(self, part, sa, lcp, bwt, minl=20, minn=2, maxlcp=None, trace=False):
        assert rank == 0 or (maxlcp == 77 and not self.constructed)
        self.front = (part, arr(sa, np.int32).copy(), arr(lcp, np.int32).copy(), arr(bwt, np.uint8).copy())
    def align_builtin_continue(self, stop):
        return len(self.sizes)
    def align_builtin_resume(self):
        part, sa, lcp, bwt = self.front
        at = 0
        import time; time.sleep(0.02 * len(part["meta"]))      # (a batch takes a while: the other ranks get to ask meanwhile)
        for k in range(len(part["meta"])):
            n = int(part["meta"][k, 1])
            assert (lcp[at:at + n] == 7).all() and (bwt[at:at + n] == 65).all()
            assert part["nodes"][part["node_first"][k], 1] - part["nodes"][part["node_first"][k], 0] == n
            self.done.append((n, (int(sa[at:at + n].sum()), int(part["nodes"][part["node_first"][k], 0]))))
            at += n
        l = np.array([d[0] for d in self.done], np.uint32); pos = np.array([p for d in self.done for p in d[1]], np.int64)
        st = shard.empty_result()["stats"]; st["splits"] = len(self.done)
        self.done = []
        return dict(stats=st, anchors=(l, np.arange(0, 2 * len(l) + 1, 2), pos), trace=None)

res = shard.align_sharded_group(FakeIndex(), grp, mem, 20, 2, stop_subs=4, per_rank=2)
if rank == 0:
    l, off, pos = res["anchors"]
    print(json.dumps({"anchors": sorted((int(l[k]), [int(x) for x in pos[off[k]:off[k + 1]]]) for k in range(len(l))), "shares": res["shares"], "batches": res["batches"], "splits": res["stats"]["splits"]}))
else:
    assert res is None
grp.barrier(); grp.close()
'''


@pytest.mark.parametrize("world,mem", [(2, "shared"), (3, "shared"), (3, "host")])
def test_divided_alignment_over_the_socket_transport(tmp_path, world, mem):
    """shard.align_sharded_group: the same work queue without torch -- requests, metadata and results over local sockets
    (reveal_amd/transport.py), the segments either in memory every rank can open (`shared`: the code path of the GPU ranks -- staging
    buffers exported once, a worker copies its batches out of them at the offsets it is told -- with POSIX shared memory in place of
    HIP's inter-process handles) or inside the messages (`host`)"""
    script = tmp_path / "group_worker.py"
    script.write_text(GROUP_WORKER % ROOT)
    port = str(29700 + world * 3 + (1 if mem == "host" else 0))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SHARD_MEM=mem)
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    r = json.loads([x for x in outs[0][0].splitlines() if x.startswith("{")][-1])
    sizes = [5, 100, 7, 40, 40, 1, 60]
    SA = np.arange(sum(sizes)) * 3 + 1
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    want = sorted([(9, [0, 500])] + [(n, [int(SA[o:o + n].sum()), int(o)]) for o, n in zip(off, sizes)])
    assert [tuple(a) for a in r["anchors"]] == [(l, p) for l, p in want]
    assert r["splits"] == 8 and sum(r["shares"]) == sum(sizes) and len(r["shares"]) == world
    nbatches = len(shard.make_batches(sizes, world, 2))
    assert sum(r["batches"]) == nbatches and sum(1 for x in r["shares"] if x > 0) >= 2


def test_queue_batches():
    sizes = [5, 100, 7, 40, 40, 1, 60]
    b = shard.make_batches(sizes, 3, 2)
    assert sorted(int(x) for p in b for x in p) == list(range(7))
    assert [int(np.asarray(sizes)[p].sum()) for p in b] == [100, 60, 80, 13]      # target 253 // 6 = 42: 100 | 60 | 40 + 40 | the rest
    assert shard.make_batches([], 4) == [] and len(shard.make_batches([9], 4)) == 1


def test_bench_gpus_flag_spawns_the_ranks():
    """`bench.py --gpus N` outside a launcher starts N ranks itself (torch.distributed.run, 127.0.0.1) and reports n_gpus = N;
    under a launcher whose world size differs from --gpus it refuses instead of silently reporting the wrong n_gpus
    (RV_BENCH_DRYRUN: the ranks meet over gloo, count themselves and stop in front of the first GPU call)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RV_BENCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert r == {"dry_run": True, "n_gpus": 2, "ranks_counted": 2, "gpus_flag": 2}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="3", RANK="0"), timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, env=env, timeout=120)
    assert json.loads(one.stdout.splitlines()[-1])["n_gpus"] == 1


@pytest.mark.parametrize("names", [("1a", "1b"), ("1e", "1b"), ("1a", "1b", "1c", "1d", "1e"), ("d1", "d2")])
def test_gfa_paths_spell_the_inputs(tmp_path, names):
    """the invariant of the reference's test15 (reveal/tests/test_reveal.py:150-159) for reveal_amd/gfa.py: the graph
    built from a run's anchors (here: the oracle's, no GPU) spells every input sequence along its path"""
    from helpers import assemble, fa, oracle
    from reveal_amd import gfa, rem
    inputs = fa(*names)
    T, nsep, nodes = assemble(inputs)
    O = oracle(False)
    r = O.align_bench(O.construct(T, nsep, len(inputs)), nodes, 20, 2)
    l, n, off, pos = r["anchors"]
    anchors = [(int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l))]
    records = [(name, s) for f in inputs for name, s in rem.fasta_reader(f)]
    seqs = [(name, iv) for (name, _), iv in zip(records, nodes)]
    segments, links, paths = gfa.build_graph(T, seqs, anchors)
    assert len(paths) == len(records)
    shared = sum(1 for k in range(len(segments)) if sum(ids.count(k + 1) for _, ids in paths) > 1)
    assert shared == len(anchors)                              # every anchor is one node on more than one path
    fn = gfa.write_gfa(str(tmp_path / "out.gfa"), segments, links, paths)
    seg, lk, pp = gfa.read_gfa(fn)
    assert [name for name, _ in pp] == [name for name, _ in records]
    ls = set(lk)
    for (name, ids), (_, s) in zip(pp, records):
        assert all((u, v) in ls for u, v in zip(ids, ids[1:]))
        assert "".join(seg[i] for i in ids) == s.upper()
    # gzip output and the default extension
    fn2 = gfa.write_gfa(str(tmp_path / "out2"), segments, links, paths)
    assert fn2.endswith(".gfa.gz") and gfa.read_gfa(fn2)[2] == pp


def test_result_arrays_own_their_pages():
    """what rv_set_result_buffers page-locks begins on a page boundary and shares no page with another object (arrays from the C heap
    did: GPU memory faults of later copies, profiles/r04_stress_fault.txt); a view keeps the array -- and its refcount -- alive"""
    import mmap
    import sys
    from reveal_amd._index import _page_array
    a, b = _page_array(3, np.uint32), _page_array(1 << 20, np.int64)
    for x in (a, b):
        assert x.ctypes.data % mmap.PAGESIZE == 0 and x.flags.writeable
    assert len(a) == 3 and len(b) == 1 << 20 and a.dtype == np.uint32
    lo, hi = sorted((a.ctypes.data, b.ctypes.data))
    assert hi - lo >= mmap.PAGESIZE
    before = sys.getrefcount(a)
    v = a[:2]
    assert sys.getrefcount(a) == before + 1 and v.base is a
    b[-1] = 7
    assert b[-1] == 7
