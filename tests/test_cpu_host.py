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
from reveal_amd import rem


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


WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from reveal_amd import synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
seqs = synth.genomes(20000, 2, seed=42 + 1000 * rank)          # bench.py's per-rank shard: its own genome pair
bases = sum(len(s) for s in seqs)
elapsed = 0.5 + rank                                           # pretend timings
t = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
b = torch.tensor([float(bases)], dtype=torch.float64); dist.all_reduce(b, op=dist.ReduceOp.SUM)
import hashlib
h = torch.tensor([int(hashlib.sha256(seqs[0]).hexdigest()[:12], 16)], dtype=torch.int64)
hs = [torch.zeros_like(h) for _ in range(world)]; dist.all_gather(hs, h)
if rank == 0:
    print(json.dumps({"tmax": t.item(), "bases": b.item(), "distinct_inputs": len({int(x.item()) for x in hs})}))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_sharding_gloo(tmp_path):
    """N>1 plumbing of bench.py on CPU: ranks take disjoint inputs (different seeds), no data-path
    collective, value = sum of bases / max time"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("{")][-1]
    r = json.loads(line)
    assert r["tmax"] == 1.5 and r["bases"] == 80000.0 and r["distinct_inputs"] == 2
