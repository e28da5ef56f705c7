"""rv_batch_run (include/reveal_amd.h): independent alignments on one GPU whose anchor cascades run their level loops as ONE set of launches
(rv_cascade_multi.hip, batch_joint_phase).  Every job's anchors, statistics and final text must be what `index.align_builtin` gives the same input alone
(which tests/test_gpu_cascade.py and the golden vectors pin against the oracle)."""
import numpy as np
import pytest

from helpers import feed, synth
from reveal_amd import batch
from test_gpu_align import mod

pytestmark = pytest.mark.gpu


def aset(r):
    l, off, pos = r["anchors"]
    return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))


def alone(M, seqs, minl=20, minn=2):
    idx = feed(M.index(), seqs)
    idx.construct()
    r = idx.align_builtin(minl, minn)
    return aset(r), {k: r["stats"][k] for k in ("steps", "splits", "anchored_bp", "levels", "maxdepth")}, idx.T, idx.cascade_info()


def family(L, K, seed, indelfrac=0.0):
    return [g.decode() for g in synth.genomes(L, K, seed=seed, indelfrac=indelfrac)]


@pytest.mark.parametrize("sa64", [False, True])
def test_jobs_of_a_batch_equal_the_jobs_alone(sa64):
    M = mod(sa64)
    inputs = [family(60000, 5, 11), family(45000, 5, 12, indelfrac=0.2), family(70000, 5, 13), family(30000, 5, 14, indelfrac=0.2)]
    want = [alone(M, s) for s in inputs]
    assert all(w[3]["done"] for w in want)                      # (the cascade decides these runs: that is the path a batch shares)
    idxs = [feed(M.index(), s) for s in inputs]
    B = batch.Batch(idxs)
    for rep in range(2):                                        # (handles and the group's buffers are reused from run to run)
        res = B.run(20, 2)
        for r, ix, w in zip(res, idxs, want):
            assert aset(r) == w[0]
            assert {k: r["stats"][k] for k in w[1]} == w[1]
            assert ix.T == w[2]
        info = B.info()
        assert info["joint_level_loops"] == rep + 1 and info["jobs_served"] == 4 * (rep + 1)
    B.close()


def test_a_mixed_batch_falls_back_job_by_job():
    """different sample counts in one batch: the joint loop refuses, every job runs its own; a two-sample job never takes part; results as alone"""
    M = mod(False)
    inputs = [family(50000, 5, 21), family(50000, 3, 22), family(80000, 2, 23), family(40000, 5, 24)]
    want = [alone(M, s) for s in inputs]
    idxs = [feed(M.index(), s) for s in inputs]
    B = batch.Batch(idxs)
    res = B.run(20, 2)
    for r, ix, w in zip(res, idxs, want):
        assert aset(r) == w[0] and ix.T == w[2]
    assert B.info()["joint_level_loops"] == 0
    B.close()


def test_a_job_that_leaves_early_does_not_hold_the_others():
    """one job's cascade gives up before the rendezvous (unrelated genomes: no match of all samples at the top), one runs through the level pipeline by
    request (RV_NO_CASCADE on its handle): the other two still meet and share their loop"""
    M = mod(False)
    unrelated = [synth.genomes(40000, 1, seed=100 + k)[0].decode() for k in range(3)]
    inputs = [family(50000, 3, 31), unrelated, family(60000, 3, 32), family(40000, 3, 33)]
    want = [alone(M, s) for s in inputs]
    idxs = [feed(M.index(), s) for s in inputs]
    idxs[3].set_option("RV_NO_CASCADE", 1)
    B = batch.Batch(idxs)
    res = B.run(20, 2)
    for r, ix, w in zip(res, idxs, want):
        assert aset(r) == w[0] and ix.T == w[2]
    assert B.info() == {"joint_level_loops": 1, "jobs_served": 2}
    B.close()


def test_batch_with_a_job_whose_sub_indices_stay_undecided():
    """repeats inside the samples leave undecided sub-indices: the joint loop hands each job its own, which it rebuilds and finishes in the level pipeline"""
    M = mod(False)
    inputs = [[g.decode() for g in synth.family(120000, 4, seed=41 + j, repeats=0.03, nruns=2)] for j in range(3)]
    want = [alone(M, s) for s in inputs]
    idxs = [feed(M.index(), s) for s in inputs]
    B = batch.Batch(idxs)
    res = B.run(20, 2)
    for r, ix, w in zip(res, idxs, want):
        assert aset(r) == w[0] and ix.T == w[2]
        assert {k: r["stats"][k] for k in ("splits", "anchored_bp")} == {k: w[1][k] for k in ("splits", "anchored_bp")}
    B.close()
