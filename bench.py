#!/usr/bin/env python3
"""bench.py -- Mbp/s indexed + MUM-anchored (reveal rem hot path) on N MI355X.

One step = construct() (suffix array, inverse, LCP) + the full recursive
anchoring (scan -> pick -> label/split/bubble per level) of one batch of
synthetic genomes whose text is already resident in HBM, with the deterministic
benchmark callbacks of SURVEY.md 8(d).  Default workload = BASELINE.json
configs[1]: 2 x 5 Mbp, 1 % SNP, -m 20.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank anchors its own genome pair (different
seed), no collective on the data path (the path shards by independent inputs);
value = bases of all ranks / max-over-ranks time.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def build_index(seqs, sa64=False):
    from reveal_amd import reveallib, reveallib64
    idx = (reveallib64 if sa64 else reveallib).index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        idx.addsequence(s)
    idx.upload()               # text resident in HBM before anything is timed
    return idx


def cpu_baseline(seqs, minl, minn):
    """the CPU restatement (oracle/, validated against the reference's own C) on this host, 1 thread"""
    from oracle import oracle_ctypes
    O = oracle_ctypes.Oracle(False)
    T, nsep, nodes = bytearray(), [], []
    for k, s in enumerate(seqs):
        if k:
            nsep.append(len(T) - 1)
        b = len(T)
        T += s + b"$"
        nodes.append((b, len(T) - 1))
    t0 = time.perf_counter()
    c = O.construct(bytes(T), nsep, len(seqs))
    t1 = time.perf_counter()
    r = O.align_bench(c, nodes, minl, minn)
    t2 = time.perf_counter()
    return dict(t_construct=t1 - t0, t_align=t2 - t1, result=r, ref_divsufsort=O.ref_divsufsort)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--L", type=int, default=5_000_000, help="genome length (bp)")
    ap.add_argument("--genomes", type=int, default=2)
    ap.add_argument("--minl", type=int, default=20)
    ap.add_argument("--minn", type=int, default=2)
    ap.add_argument("--sa64", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--prof-all", action="store_true", help="time every kernel class inside the timed region (adds events to every level)")
    ap.add_argument("--divide", action="store_true",
                    help="N>1: ONE alignment divided over the ranks (frontier hand-off, reveal_amd/shard.py; strong scaling) "
                         "instead of one alignment per rank")
    ap.add_argument("--jobs", type=int, default=1,
                    help="independent alignments per rank, run concurrently (one index handle, HIP stream and host thread each): "
                         "config 5's 20 independent 5-genome jobs on 8 GPUs (reveal/align.py:45-53); the default 1 is the metric's config")
    ap.add_argument("--cpu-L", type=int, default=0, help="genome length for the CPU sample (default: same as --L, capped at 5 Mbp)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("RV_BENCH_SHARE_GPU"):      # plumbing check on a one-GPU box: every rank on device 0, gloo instead of RCCL
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from reveal_amd import _lib, synth
    _lib.set_device(local_rank)

    divide = args.divide and world > 1
    jobs = max(1, args.jobs) if not divide else 1
    seqs = synth.genomes(args.L, args.genomes, seed=42 + (0 if divide else 1000 * rank))
    bases = sum(len(s) for s in seqs)
    idx = build_index(seqs, args.sa64)
    # further jobs of this rank: their own inputs (other seeds), handles and streams
    extra = [build_index(synth.genomes(args.L, args.genomes, seed=42 + 1000 * rank + 17 * j), args.sa64) for j in range(1, jobs)]
    bases *= jobs

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if divide:      # rank 0 constructs and runs the top levels, every rank finishes a share of the frontier
            from reveal_amd import shard
            return shard.align_sharded(idx, args.minl, args.minn)
        if extra:      # (ctypes releases the GIL inside the library calls: the jobs' level loops overlap on the GPU)
            import threading
            res = [None] * (1 + len(extra))

            def run(k, ix):
                ix.construct()
                res[k] = ix.align_builtin(args.minl, args.minn)
            th = [threading.Thread(target=run, args=(k + 1, ix)) for k, ix in enumerate(extra)]
            for t in th:
                t.start()
            run(0, idx)
            for t in th:
                t.join()
            if any(r is None for r in res):
                raise RuntimeError("a concurrent job failed")
            return res[0]
        idx.construct()
        return idx.align_builtin(args.minl, args.minn)

    for _ in range(args.warmup):
        step()
    # inside the timed region only the roofline-judged kernel is timed (two HIP events per launch on the library's stream);
    # the per-class breakdown comes from two extra, untimed steps with every class timed
    kname = "scan_multi" if args.genomes > 2 else "scan_pair"
    for ix in [idx] + extra:
        ix.prof(enable=True, reset=True, only=None if args.prof_all else (kname,))
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = idx.prof(enable=False)
    for ix in extra:      # the judged kernel over all jobs (launches that overlap other jobs' kernels share the GPU with them)
        p2 = ix.prof(enable=False)
        prof = {k: tuple(a + b for a, b in zip(prof[k], p2[k])) for k in prof}
    breakdown = None
    if rank == 0 and not divide:
        idx.prof(enable=True, reset=True)
        for _ in range(2):
            idx.construct()
            idx.align_builtin(args.minl, args.minn)
        breakdown = {k: v[1] / 2 for k, v in idx.prof(enable=False).items() if v[0]}
    barrier()

    total_bases, tmax = bases, elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        b = torch.tensor([bases], dtype=torch.float64, device="cuda")
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        tmax, total_bases = float(t.item()), float(b.item())
        if divide:
            total_bases = float(bases)      # every rank worked on the same inputs

    if rank == 0:
        launches, ms, nbytes = prof[kname]
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_scan.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                # per-launch HBM bytes from the PMC passes only describe the workload they were collected on
                if pj.get("workload") == "%dx%d-%d" % (args.genomes, args.L, 64 if args.sa64 else 32):
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        st = last["stats"]
        out = {
            "metric": "Mbp/s indexed+MUM-anchored (reveal rem)",
            "value": total_bases * args.steps / tmax / 1e6,
            "unit": "Mbp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if divide else "weak",
            "vs_baseline": None,
            "dtype": "int64" if args.sa64 else "int32",
            "data": "synthetic",
            "config": {"workload": "%dx synthetic %g Mbp genomes (uniform ACGT, 1%% SNP, seed 42+1000*rank), rem -m %d -n %d, "
                                   "construct + full recursion, bench picker" % (args.genomes, args.L / 1e6, args.minl, args.minn),
                       "bases_per_gpu": bases, "index": "64-bit" if args.sa64 else "32-bit",
                       "jobs_per_gpu": jobs,
                       "sharding": ("one alignment divided over %d ranks (frontier hand-off), shares %s" % (world, last.get("shares"))) if divide
                                   else ("one alignment per rank, no exchange" if jobs == 1 else
                                         "%d independent alignments per rank, concurrently (a handle, stream and host thread each), no exchange" % jobs)},
            "roofline": {"bound": "hbm", "kernel": "k_scan_" + ("multi" if args.genomes > 2 else "pair"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launches": launches, "avg_us": (ms * 1e3 / launches) if launches else None,
                         "algorithmic_bytes_per_launch": (nbytes / launches) if launches else None},
            "breakdown_ms_per_step": breakdown,
            "recursion": {"anchors": st["splits"], "anchored_bp": st["anchored_bp"], "levels": st["levels"], "subindices": st["steps"],
                          "scanned_ranks": st["scanned_ranks"], "host_s": st["t_host"], "scan_s": st["t_scan"],
                          "split_s": st["t_split"], "bubble_s": st["t_bubble"]},
            "sa_build": idx.sa_stats(),
        }
        if world == 1 and not args.no_cpu:
            cl = args.cpu_L or min(args.L, 5_000_000)
            cseqs = seqs if cl == args.L else synth.genomes(cl, args.genomes, seed=42)
            cb = cpu_baseline(cseqs, args.minl, args.minn)
            cbases = sum(len(s) for s in cseqs)
            out["cpu_baseline"] = {
                "value": cbases / (cb["t_construct"] + cb["t_align"]) / 1e6, "unit": "Mbp/s", "cores": 1, "kind": "port",
                "sample": "%dx %g Mbp (same generator), construct %.2f s + recursion %.2f s, single thread, SA by %s" % (
                    args.genomes, cl / 1e6, cb["t_construct"], cb["t_align"],
                    "the reference's divsufsort (oracle/_ref)" if cb["ref_divsufsort"] else "the oracle's own sorter"),
                "host_cores_available": os.cpu_count(),
            }
            if cl == args.L:          # free full-size parity check: same anchors as the CPU path
                rl, rn, roff, rpos = cb["result"]["anchors"]
                gl, goff, gpos = last["anchors"]
                ra = sorted((int(rl[k]), tuple(int(x) for x in rpos[roff[k]:roff[k + 1]])) for k in range(len(rl)))
                ga = sorted((int(gl[k]), tuple(int(x) for x in gpos[goff[k]:goff[k + 1]])) for k in range(len(gl)))
                out["parity"] = {"anchors_gpu": len(ga), "anchors_cpu": len(ra), "identical_anchor_set": ra == ga,
                                 "identical_final_text": idx.T.encode("latin-1") == cb["result"]["T"]}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
