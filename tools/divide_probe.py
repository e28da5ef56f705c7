"""Where would the time of ONE divided alignment go?  (reveal_amd/shard.py, SURVEY.md 8(e) second granularity)

Emulates an N-rank divided run on ONE GPU by running the ranks' shares one after the other, and prints the
critical path an N-GPU run would have: owner (construct + top levels + packing) + the slowest share.  This is a
projection from single-GPU timings -- it leaves out the xGMI transfer (printed as bytes) -- not a multi-GPU measurement.

usage: python tools/divide_probe.py [L=5000000] [ranks=8] [stop_subs=8*ranks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from reveal_amd import reveallib, shard, synth


def feed(idx, seqs):
    for g in seqs:
        idx.addsample("s")
        idx.addsequence(g.decode())
    return idx


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    stop = int(sys.argv[3]) if len(sys.argv) > 3 else 8 * ranks
    seqs = synth.genomes(L, 2, seed=42)
    sync = torch.cuda.synchronize
    one = feed(reveallib.index(), seqs)
    one.upload()
    for _ in range(2):      # warm
        one.construct(); ref = one.align_builtin(20, 2)
    sync(); t0 = time.perf_counter()
    one.construct(); sync(); t1 = time.perf_counter()
    ref = one.align_builtin(20, 2); sync(); t2 = time.perf_counter()
    undiv = (t2 - t0) * 1e3
    print("undivided: construct %.1f ms + recursion %.1f ms = %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, undiv))
    owner = one
    workers = [feed(reveallib.index(), seqs) for _ in range(ranks - 1)]
    for w in workers:
        w.upload()
    best = None
    for it in range(3):
        sync(); t0 = time.perf_counter()
        owner.construct(); sync(); t1 = time.perf_counter()
        left = owner.align_builtin_until(stop, 20, 2); sync(); t2 = time.perf_counter()
        if left == 0:
            print("the run finished before the frontier reached %d sub-indices" % stop)
            return
        fr = owner.frontier()
        parts = shard.partition(fr["meta"][:, 1], ranks)
        packed = []
        for p in parts:
            part = shard.subset(fr, p)
            m = int(part["meta"][:, 1].sum())
            bufs = shard._buffers(owner._lib, m, "cuda:0")
            owner.frontier_pack(p, *bufs)
            packed.append((part, bufs, m))
        sync(); t3 = time.perf_counter()
        owner.frontier_import(packed[0][0], *packed[0][1], minl=20, minn=2)
        res = [owner.align_builtin_resume()]
        sync(); t4 = time.perf_counter()
        share_ms = [(t4 - t3) * 1e3]
        for w, (part, bufs, m) in zip(workers, packed[1:]):
            sync(); a = time.perf_counter()
            if m:
                w.frontier_import(part, *bufs, minl=20, minn=2, maxlcp=owner.maxlcp)
                res.append(w.align_builtin_resume())
            sync(); share_ms.append((time.perf_counter() - a) * 1e3)
        got = shard.merge(res)
        assert got["stats"]["splits"] == ref["stats"]["splits"] and got["stats"]["anchored_bp"] == ref["stats"]["anchored_bp"]
        crit = (t3 - t0) * 1e3 + max(share_ms)
        line = dict(level=fr["level"], subs=left, construct=(t1 - t0) * 1e3, top_levels=(t2 - t1) * 1e3, pack=(t3 - t2) * 1e3,
                    shares=[pk[2] for pk in packed], share_ms=share_ms, critical=crit)
        if best is None or crit < best["critical"]:
            best = line
    b = best
    moved = sum(b["shares"][1:]) * 9
    print("divided over %d ranks at level %d (%d sub-indices): construct %.1f + top levels %.1f + pack %.1f ms on the owner" % (
        ranks, b["level"], b["subs"], b["construct"], b["top_levels"], b["pack"]))
    print("  shares (ranks): %s" % b["shares"])
    print("  shares (ms, import + resume): %s" % ["%.1f" % x for x in b["share_ms"]])
    print("  segments leaving the owner: %.1f MB (%.2f ms at 50 GB/s per link, links in parallel: %.2f ms)" % (
        moved / 1e6, moved / 50e9 * 1e3, max(b["shares"][1:]) * 9 / 50e9 * 1e3))
    print("  projected critical path %.1f ms vs %.1f ms undivided -> x%.2f (Amdahl bound from the owner part: x%.2f)" % (
        b["critical"], undiv, undiv / b["critical"], undiv / (b["construct"] + b["top_levels"] + b["pack"])))


if __name__ == "__main__":
    main()
