#!/usr/bin/env python3
"""bench.py -- Mbp/s indexed + MUM-anchored (reveal rem hot path) on N MI355X.

One step = construct() (suffix array, LCP, BWT) + the full recursive anchoring
of one batch of synthetic genomes whose text is ALREADY RESIDENT IN HBM when the
timed region starts (the host->device copy of the text is outside it), with the
deterministic benchmark callbacks of SURVEY.md 8(d).  Two samples: the anchor
cascade (rv_cascade.hip) decides the recursion from the top-level scan and
rebuilds what it cannot decide; more samples: scan -> pick -> label / split /
bubble per level.  Either way the anchors are the reference recursion's.

Default workload = BASELINE.json configs[3], the largest single-GPU
configuration: 2 x 250 Mbp, 1 % SNP, -m 20 (n = 5*10^8, 32-bit index).
`--L 5000000` gives configs[1] (2 x 5 Mbp), `--L 5000000 --genomes 10` configs[2].

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
torch.distributed.run with N ranks (one process per GPU); under a launcher whose
WORLD_SIZE differs from --gpus it refuses to run.

N > 1, one process per GPU, no collective on the data path in either mode
(independent `reveal rem` jobs are the reference's only parallelism, reveal/align.py:27-54):
  * per-rank ("weak"): every rank anchors its own genome set -- the line's `value`;
  * divide ("strong"): ONE alignment divided over the ranks -- rank 0 constructs and
    runs the top levels, the ranks pull batches of the frontier's sub-indices from it
    (reveal_amd/shard.py; BASELINE config 4's "interval-split scaling") -- reported in
    the same line under "divide" for inputs of at least 100 Mbp (the default workload).
  --mode per-rank / --mode divide run one of the two only.
value = bases of the whole job / max-over-ranks time.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CPU_SAMPLE_L = 20_000_000   # the CPU legs and the parity leg run on a 2 x 20 Mbp sample of the same generator (about 11 s of single-thread CPU work)


def build_index(seqs, sa64=False):
    from reveal_amd import reveallib, reveallib64
    idx = (reveallib64 if sa64 else reveallib).index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k)
        for c in (s if isinstance(s, list) else [s]):      # (a sample may be a list of contigs: one addsequence each, utils.py:325-350)
            idx.addsequence(c)
    import torch
    idx.upload()               # text resident in HBM before anything is timed (this first copy also pays for the handle's device allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx.upload_again()         # the copy alone
    torch.cuda.synchronize()
    idx.upload_ms = (time.perf_counter() - t0) * 1e3
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


def cpu_all_cores(L, genomes, minl, minn, max_workers=0):
    """All host cores: P independent single-threaded alignments at once (the reference's only parallelism is independent
    `reveal rem` jobs, reveal/align.py:45-53; its C path is single-threaded).  P = the cores this process may use, capped by
    free memory (~0.9 GB per 2 x 5 Mbp job, ~3 GB per 2 x 20 Mbp).  Workers are separate processes (oracle/cpu_worker.py) released at a common
    start time; rate = all bases / (latest end - earliest start)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:      # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) bounds what "all cores" can mean here
        q = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            q = None if a == "max" else float(a) / float(b)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            a = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); b = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            q = a / b if a > 0 else None
        if q:
            cores = max(1, min(cores, int(q + 0.5)))
    except Exception:
        pass
    P = min(cores, 64)      # bounded sample: at most 64 jobs at once (a host with more cores is stated as such in the line)
    try:
        import psutil
        per_job = 0.07e9 * genomes * (L / 1e6) + 0.2e9      # ~0.9 GB for 2 x 5 Mbp incl. the interpreter
        P = max(1, min(P, int(psutil.virtual_memory().available * 0.6 / per_job)))
    except Exception:
        pass
    if max_workers:
        P = min(P, max_workers)
    start_at = time.time() + max(6.0, 0.06 * P)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), "--L", str(L), "--genomes", str(genomes),
                               "--seed", str(5000 + 13 * k), "--minl", str(minl), "--minn", str(minn), "--start-at", repr(start_at)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS="1")) for k in range(P)]
    res = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode == 0:
            try:
                res.append(json.loads(out.decode().strip().splitlines()[-1]))
            except Exception:
                pass
    if not res:
        return None
    span = max(r["t_end"] for r in res) - min(r["t_begin"] for r in res)
    return dict(workers=len(res), launched=P, cores=cores, bases=sum(r["bases"] for r in res), seconds=span,
                late=max(r["late"] for r in res), value=sum(r["bases"] for r in res) / span / 1e6)


def spawn_ranks(n):
    """`bench.py --gpus N` outside a launcher: the same command line again under torch.distributed.run, one rank per GPU"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def run_config5(args, rank, local_rank, world, dist, torch):
    """BASELINE config 5, level 0: `reveal align --order=sequential --chunksize=5` over 100 genomes of 5 Mbp prints 20 independent
    `reveal rem` jobs of five FASTA inputs each (reveal/align.py:27-54; reveal_amd/align.py sequential_plan) -- job j belongs to rank
    j % world, a rank runs `--jobs` of its jobs at a time (a handle, HIP stream and host thread each), nothing is exchanged.  One step =
    every level-0 job once.  Levels 1-2 align the level-0 GRAPHS (GFA -> GFA through the Python graph callbacks, reveal_amd/rem.py
    graph_rem): the Python driver's work, outside the timed scope as SURVEY 8(e) allows."""
    import queue
    import threading
    import numpy as np
    from reveal_amd import _lib, align, check, synth
    _lib.set_device(local_rank)
    NG, CH = args.c5_genomes, 5
    plan = align.sequential_plan(list(range(NG)), CH)
    level0 = [members for members, _ in plan[0]]
    mine = [j for j in range(len(level0)) if j % world == rank]
    base = synth.base_codes(args.L, 42)
    seqs = {j: [synth.member(base, k, 42, indelfrac=args.indelfrac) for k in level0[j]] for j in mine}
    handles = {j: build_index(seqs[j], args.sa64) for j in mine}
    upload_ms = sum(h.upload_ms for h in handles.values())
    nthreads = max(1, min(args.jobs if args.jobs > 1 else 8, len(mine)))      # (measured on one GPU: 2 / 4 / 8 / 12 / 20 at a time -> 114 / 99 / 95 / 96 / 103 ms per step)
    results = {}

    batches = None
    if args.batch > 0:      # rv_batch_run: the jobs of a batch on host threads inside the library, the level loops of their anchor cascades as one set of launches
        from reveal_amd import batch as rvbatch
        batches = [(grp, rvbatch.Batch([handles[j] for j in grp])) for grp in (mine[i:i + args.batch] for i in range(0, len(mine), args.batch))]

    def step():
        if batches is not None:
            for grp, B in batches:
                for j, r in zip(grp, B.run(args.minl, args.minn)):
                    results[j] = r
            return
        todo = queue.Queue()
        for j in mine:
            todo.put(j)
        err = []

        def work():
            while True:
                try:
                    j = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    handles[j].construct()
                    results[j] = handles[j].align_builtin(args.minl, args.minn)
                except Exception as e:      # noqa: BLE001  (reported below: a failed job fails the step)
                    err.append((j, e))
        th = [threading.Thread(target=work) for _ in range(nthreads - 1)]
        for t in th:
            t.start()
        work()
        for t in th:
            t.join()
        if err:
            raise RuntimeError("config-5 job %d failed: %s" % err[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for h in handles.values():
        h.prof(enable=True, reset=True, only=("scan_multi",))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = [0, 0.0, 0.0]
    for h in handles.values():
        p = h.prof(enable=False)["scan_multi"]
        prof = [prof[0] + p[0], prof[1] + p[1], prof[2] + p[2]]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # every job of this rank: the full-size properties of its last result; job 0 also against the CPU path's digests
    props, golden = {}, None
    if not args.no_check:
        for j in mine:
            T0 = np.frombuffer(b"$".join(seqs[j]) + b"$", dtype=np.uint8)
            nsep = np.cumsum([len(x) + 1 for x in seqs[j]])[:-1] - 1
            T1 = handles[j].array("T")
            props[j] = bool(check.recursion_properties(T0, T1, results[j]["anchors"], nsep, args.minl)["all"])
            if j == 0:
                rec = check.golden_record(args.L, CH, 42, args.indelfrac, args.minl, args.minn)
                if rec is not None:
                    golden = check.compare_with_golden(rec, anchors=results[j]["anchors"], T_final=T1)
    mine_info = {"rank": rank, "jobs": mine, "properties": props, "anchors": {j: int(results[j]["stats"]["splits"]) for j in mine}}
    gathered = [mine_info]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine_info)
    if rank == 0:
        total_bases = float(NG * args.L)
        launches, ms, nbytes = prof
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        allprops = {j: v for g in gathered for j, v in g["properties"].items()}
        out = {
            "metric": "Mbp/s indexed+MUM-anchored (reveal rem)", "value": total_bases * args.steps / elapsed / 1e6, "unit": "Mbp/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int64" if args.sa64 else "int32", "data": "synthetic",
            "config": {"workload": "BASELINE config 5, level 0: %d synthetic %g Mbp genomes of one family (uniform ACGT base, 1%% SNP per member, seed 42), "
                                   "--order=sequential --chunksize=%d -> %d independent rem jobs of %d inputs each (reveal/align.py:27-54), rem -m %d -n %d, "
                                   "construct + full recursion per job, bench picker; texts resident in HBM before the timed region; levels 1-2 "
                                   "(GFA -> GFA through the Python graph callbacks) are the Python driver's and outside the timed scope (SURVEY 8(e))"
                                   % (NG, args.L / 1e6, CH, len(level0), CH, args.minl, args.minn),
                       "jobs": len(level0), "jobs_per_rank": [len(g["jobs"]) for g in gathered], "concurrent_jobs_per_gpu": nthreads if batches is None else args.batch,
                       "batched": None if batches is None else {"jobs_per_rv_batch_run": args.batch, "joint_level_loops": sum(B.info()["joint_level_loops"] for _, B in batches),
                                                                "jobs_served_by_them": sum(B.info()["jobs_served"] for _, B in batches)},
                       "bases_per_step": total_bases, "index": "64-bit" if args.sa64 else "32-bit",
                       "sharding": "job j on rank j % world, no exchange (the reference's only parallelism: independent reveal rem commands)"},
            "upload_ms": upload_ms,
            "roofline": {"bound": "hbm", "kernel": "k_casm_scan / k_scan_multi (the multi-sample scans of rank 0's jobs)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": launches,
                         "avg_us": (ms * 1e3 / launches) if launches else None},
            "jobs_anchors": {str(j): v for g in gathered for j, v in g["anchors"].items()},
            "properties_full_size": {"all": all(allprops.values()) and len(allprops) == len(level0) if not args.no_check else None,
                                     "jobs_checked": len(allprops), "failed": [j for j, v in allprops.items() if not v]},
            "parity": {"full_size": golden if golden is not None else "no CPU digests for job 0 in tests/golden/fullsize.json"},
        }
        print(json.dumps(out))


def run_stream(args, rank, local_rank, world, dist, torch, emit=True):
    """--config stream: a queue of `--pairs` inputs per rank (default 20 pairs of 2 x 250 Mbp = 10 Gbp, north_star's target volume), each
    one assembled on the host (addsample / addsequence into the handle's page-locked text), copied into HBM, constructed and anchored --
    everything from the caller's sequences in ordinary host memory to the anchors in the caller's arrays is inside the timed region.
    `--stream-handles` inputs are in flight at a time, each on its own handle, HIP stream and host thread: the next input's assembly
    and its host->device copy (one DMA) run under the current input's construct.  Handles are reused through rv_reset."""
    import queue
    import threading
    import numpy as np
    from reveal_amd import _lib, check, synth, reveallib, reveallib64
    if os.environ.get("RV_BENCH_SHARE_GPU"):
        local_rank = 0
    _lib.set_device(local_rank)
    P, D, H = args.pairs, max(1, min(args.stream_distinct, args.pairs)), max(1, args.stream_handles)
    # the inputs, generated before anything is timed (a caller would have read them from its files): D distinct ones, used in turn
    inputs = [synth.genomes(args.L, args.genomes, seed=42 + 1000 * rank + 17 * d, indelfrac=args.indelfrac) for d in range(D)]
    bases = [sum(len(x) for x in seqs) for seqs in inputs]
    text_bytes = max(sum(len(x) + 1 for x in seqs) for seqs in inputs)
    M = reveallib64 if args.sa64 else reveallib
    handles = [M.index() for _ in range(H)]
    hold = [None] * H                        # per handle: (which distinct input, its result) of the last run
    anchors_of = {}
    times = {"assemble": 0.0, "upload": 0.0, "construct": 0.0, "align": 0.0}
    tl = threading.Lock()

    def one(hk, i):
        ix, seqs = handles[hk], inputs[i % D]
        t0 = time.perf_counter()
        ix.reset(reserve=text_bytes)
        for k, sq in enumerate(seqs):
            ix.addsample("g%d" % k)
            ix.addsequence(sq)
        t1 = time.perf_counter()
        ix.upload()
        t2 = time.perf_counter()
        ix.construct()
        t3 = time.perf_counter()
        hold[hk] = None                      # (the previous result of this handle is let go: its arrays are the next run's)
        r = ix.align_builtin(args.minl, args.minn)
        t4 = time.perf_counter()
        hold[hk] = (i % D, r)
        with tl:
            times["assemble"] += t1 - t0; times["upload"] += t2 - t1; times["construct"] += t3 - t2; times["align"] += t4 - t3
            anchors_of[i % D] = int(r["stats"]["splits"])

    def run(npairs):
        todo = queue.Queue()
        for i in range(npairs):
            todo.put(i)
        err = []

        def work(hk):
            while True:
                try:
                    i = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    one(hk, i)
                except Exception as e:      # noqa: BLE001
                    err.append(e)
                    return
        th = [threading.Thread(target=work, args=(hk,)) for hk in range(1, H)]
        for t in th:
            t.start()
        work(0)
        for t in th:
            t.join()
        if err:
            raise err[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        run(max(H, D) * 2)                   # every handle has made its allocations, every distinct input has run
    for k in times:
        times[k] = 0.0
    kname = "scan_pair" if args.genomes == 2 else "scan_multi"
    for ix in handles:
        ix.prof(enable=True, reset=True, only=(kname,))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run(P)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = [0, 0.0, 0.0]
    for ix in handles:
        p = ix.prof(enable=False)[kname]
        prof = [prof[0] + p[0], prof[1] + p[1], prof[2] + p[2]]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if not os.environ.get("RV_BENCH_SHARE_GPU") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the last result of every handle against the full-size properties; rank 0's seed-42 input against the CPU digests
    props, golden = [], None
    if not args.no_check:
        for hk in range(H):
            if hold[hk] is None:
                continue
            dd, r = hold[hk]
            T0 = np.frombuffer(b"$".join(inputs[dd]) + b"$", dtype=np.uint8)
            T1 = handles[hk].array("T")
            nsep = np.cumsum([len(x) + 1 for x in inputs[dd]])[:-1] - 1
            props.append(bool(check.recursion_properties(T0, T1, r["anchors"], nsep, args.minl)["all"]))
            if rank == 0 and dd == 0 and golden is None:
                rec = check.golden_record(args.L, args.genomes, 42, args.indelfrac, args.minl, args.minn)
                if rec is not None:
                    golden = check.compare_with_golden(rec, anchors=r["anchors"], T_final=T1)
            del T0, T1
    info = {"rank": rank, "properties": props}
    gathered = [info]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
    if rank == 0:
        per_step = float(sum(bases[i % D] for i in range(P)))
        total = per_step * args.steps * world
        launches, ms, nbytes = prof
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        allp = [v for g in gathered for v in g["properties"]]
        npair = P * args.steps
        out = {
            "metric": "Mbp/s indexed+MUM-anchored (reveal rem)", "value": total / elapsed / 1e6, "unit": "Mbp/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64" if args.sa64 else "int32", "data": "synthetic",
            "config": {"workload": "sustained stream: %d inputs per rank and step, each %dx synthetic %g Mbp genomes (uniform ACGT, 1%% SNP; %d distinct inputs, "
                                   "seeds 42+1000*rank+17*d, used in turn), rem -m %d -n %d, construct + full recursion, bench picker.  TIMED PER INPUT: host "
                                   "assembly (addsample / addsequence from the caller's sequences in pageable host memory), host->device copy, construct, "
                                   "recursion, anchors into the caller's arrays; %d inputs in flight per GPU (a handle, HIP stream and host thread each, "
                                   "handles reused through rv_reset)" % (P, args.genomes, args.L / 1e6, D, args.minl, args.minn, H),
                       "inputs_per_rank_and_step": P, "Gbp_per_rank_and_step": per_step / 1e9, "in_flight_per_gpu": H,
                       "index": "64-bit" if args.sa64 else "32-bit", "sharding": "an input queue per rank, no exchange"},
            "includes_upload": True,
            "wall_seconds": elapsed, "Gbp_total": total / 1e9, "ms_per_input": elapsed / npair * 1e3,
            "host_thread_ms_per_input": {k: v / npair * 1e3 for k, v in times.items()},
            "roofline": {"bound": "hbm", "kernel": "k_scan_" + ("pair" if args.genomes == 2 else "multi") + " (launches of every handle; they share the GPU with the other inputs in flight)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": launches,
                         "avg_us": (ms * 1e3 / launches) if launches else None},
            "properties_full_size": {"all": (all(allp) and len(allp) > 0) if not args.no_check else None, "results_checked": len(allp)},
            "parity": {"full_size": golden if golden is not None else "no CPU digests for the last input of a handle in tests/golden/fullsize.json"},
            "anchors_per_distinct_input": {str(d): v for d, v in sorted(anchors_of.items())},
        }
        if emit:
            print(json.dumps(out))
        return out
    return None


CLASSES = ("snp0.1", "snp1", "snp5", "snp15", "indel", "repeats", "repeats_indel", "contigs50", "unrelated", "identical")
MORE_CLASSES = ("repeats10", "tandem2", "snp25")      # beyond the table: 10 % of the base in repeat copies, 2 % in tandem arrays (homopolymers included), 25 % divergence


def class_inputs(name, L, seed=42):
    """the two genomes of one input class at length L (bench.py --classes): what the headline's generator is one point of"""
    from reveal_amd import synth
    if name.startswith("snp"):
        return synth.family(L, 2, seed=seed, snp=float(name[3:]) / 100.0)
    if name == "indel":
        return synth.family(L, 2, seed=seed, indelfrac=0.2)
    if name == "repeats":                     # interspersed repeat families + tandem arrays + runs of N (synth.overlay_repeats)
        return synth.family(L, 2, seed=seed, repeats=0.02, nruns=max(3, L // 10_000_000))
    if name == "repeats_indel":
        return synth.family(L, 2, seed=seed, indelfrac=0.2, repeats=0.02, nruns=max(3, L // 10_000_000))
    if name == "repeats10":
        return synth.family(L, 2, seed=seed, repeats=0.10, nruns=max(3, L // 10_000_000))
    if name == "tandem2":
        return synth.family(L, 2, seed=seed, tandem=0.02)
    if name == "contigs50":
        return cut_into_contigs(synth.genomes(L, 2, seed=seed), 50)
    if name == "unrelated":
        return [synth.genomes(L, 1, seed=seed)[0], synth.genomes(L, 1, seed=seed + 7777)[0]]
    if name == "identical":
        g = synth.genomes(L, 1, seed=seed)[0]
        return [g, g]
    raise ValueError(name)


def run_one_class(args):
    """one class, one process (bench.py --class-one NAME): 1 warm-up + 2 timed steps, properties, CPU digests where tests/golden/fullsize.json holds them"""
    import numpy as np
    import torch
    from reveal_amd import _lib, check
    _lib.set_device(0)
    name = args.class_one
    t0 = time.perf_counter()
    seqs = class_inputs(name, args.L)
    gen_s = time.perf_counter() - t0
    bases = sum(len(c) for c in flat(seqs))
    idx = build_index(seqs, args.sa64)
    idx.construct(); idx.align_builtin(args.minl, args.minn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = None
    for _ in range(2):
        r = None
        idx.construct()
        r = idx.align_builtin(args.minl, args.minn)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    info = idx.cascade_info()
    out = {"class": name, "L": args.L, "bases": bases, "ms_per_step": dt * 1e3, "Mbp_per_s": bases / dt / 1e6, "anchors": int(r["stats"]["splits"]),
           "anchored_bp": int(r["stats"]["anchored_bp"]), "levels": int(r["stats"]["levels"]), "maxlcp": idx.maxlcp,
           "path": "cascade" if info["done"] else "level pipeline", "cascade_why": info["why"], "cascade": info, "sa_build": idx.sa_stats(), "generate_s": gen_s}
    if not args.no_check:
        T0 = np.frombuffer(b"$".join(flat(seqs)) + b"$", dtype=np.uint8)
        nsep = np.asarray(sample_seps(seqs), dtype=np.int64)
        T1 = idx.array("T")
        out["properties"] = bool(check.recursion_properties(T0, T1, r["anchors"], nsep, args.minl, collinear=name != "contigs50")["all"])
        kw = {"snp0.1": dict(snp=0.001), "snp1": {}, "snp5": dict(snp=0.05), "snp15": dict(snp=0.15), "indel": dict(indelfrac=0.2),
              "repeats": dict(repeats=0.02, nruns=max(3, args.L // 10_000_000)), "repeats_indel": dict(indelfrac=0.2, repeats=0.02, nruns=max(3, args.L // 10_000_000))}.get(name)
        rec = check.golden_record(args.L, 2, 42, minl=args.minl, minn=args.minn, **kw) if kw is not None else None
        out["golden"] = check.compare_with_golden(rec, anchors=r["anchors"], T_final=T1) if rec is not None else None
    print(json.dumps(out))


def run_classes(args):
    """bench.py --classes: every input class at the workload's size, each in a process of its own under a time limit"""
    names = [c for c in (args.classes.split(",") if args.classes != "all" else CLASSES)]
    rows = []
    for name in names:
        cmd = [sys.executable, os.path.abspath(__file__), "--class-one", name, "--L", str(args.L), "--minl", str(args.minl), "--minn", str(args.minn)] + \
              (["--sa64"] if args.sa64 else []) + (["--no-check"] if args.no_check else [])
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.class_timeout)
            line = [x for x in p.stdout.decode().splitlines() if x.startswith("{")]
            row = json.loads(line[-1]) if p.returncode == 0 and line else {"class": name, "failed": p.stderr.decode()[-400:]}
        except subprocess.TimeoutExpired:
            row = {"class": name, "failed": "no result within %d s" % args.class_timeout}
        row["wall_s"] = time.perf_counter() - t0
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    head = next((r for r in rows if r.get("class") == "snp1" and "ms_per_step" in r), None)
    for r in rows:
        if head and "ms_per_step" in r:
            r["vs_headline"] = r["ms_per_step"] / head["ms_per_step"]
    print(json.dumps({"what": "input classes at 2 x %g Mbp, construct + full recursion with the bench picker, one MI355X; snp1 = the headline's generator" % (args.L / 1e6),
                      "L": args.L, "classes": rows}))


def cut_into_contigs(seqs, k, seed=7):
    """--contigs K: every genome as K contigs cut at seeded random positions, the second and later samples' contigs in another order
    (a draft assembly against a draft assembly)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for j, s in enumerate(seqs):
        at = np.sort(rng.choice(np.arange(1000, len(s) - 1000), size=k - 1, replace=False)).tolist()
        parts = [s[a:b] for a, b in zip([0] + at, at + [len(s)])]
        if j:
            parts = [parts[i] for i in rng.permutation(len(parts))]
        out.append(parts)
    return out


def flat(seqs):
    return [c for s in seqs for c in (s if isinstance(s, list) else [s])]


def sample_seps(seqs):
    """text position of the last '$' of every sample but the last (nsep, interface.c:18-49)"""
    pos, out = 0, []
    for s in seqs:
        pos += sum(len(c) + 1 for c in (s if isinstance(s, list) else [s]))
        out.append(pos - 1)
    return out[:-1]


def anchor_set(l, off, pos):
    return sorted((int(l[k]), tuple(int(x) for x in pos[off[k]:off[k + 1]])) for k in range(len(l)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--L", type=int, default=250_000_000, help="genome length (bp)")
    ap.add_argument("--genomes", type=int, default=2)
    ap.add_argument("--minl", type=int, default=20)
    ap.add_argument("--minn", type=int, default=2)
    ap.add_argument("--sa64", action="store_true")
    ap.add_argument("--indelfrac", type=float, default=0.0,
                    help="variants by the reference's own mutation model (utils/simulate.py:17-77: this fraction of the 1 %% events are indels, "
                         "zipf(1.7) lengths) instead of substitutions only; 0 = SURVEY 8(d)'s generator, the metric's workload")
    ap.add_argument("--snp", type=float, default=0.01, help="substitution rate of the variants (0.01 = the metric's workload)")
    ap.add_argument("--repeats", type=float, default=0.0, help="fraction of the base covered by interspersed repeat copies (synth.family; implies --no-cpu --no-extra)")
    ap.add_argument("--nruns", type=int, default=0, help="runs of N in every genome (synth.family)")
    ap.add_argument("--contigs", type=int, default=1, help="every genome as this many contigs (a multi-sequence FASTA per sample); implies --no-cpu --no-extra")
    ap.add_argument("--no-extra", action="store_true", help="skip the companion legs of the default line (level pipeline, indel workload)")
    ap.add_argument("--config", choices=("default", "c5", "stream"), default="default",
                    help="c5 = BASELINE config 5's level 0: 100 genomes of 5 Mbp as 20 independent jobs of five, divided over the ranks; "
                         "stream = a queue of --pairs inputs per rank, host assembly and upload inside the timed region, overlapped with compute")
    ap.add_argument("--classes", default="", help="'all' or a comma-separated list of %s: the input-class table at --L (each class in its own process)" % (CLASSES,))
    ap.add_argument("--class-one", default="", help="(used by --classes) run one class in this process")
    ap.add_argument("--class-timeout", type=int, default=600)
    ap.add_argument("--pairs", type=int, default=20, help="--config stream: inputs per rank and step (20 x 2 x 250 Mbp = 10 Gbp)")
    ap.add_argument("--stream-distinct", type=int, default=3, help="--config stream: distinct synthetic inputs generated (used in turn)")
    ap.add_argument("--stream-handles", type=int, default=4, help="--config stream: inputs in flight per GPU")
    ap.add_argument("--c5-genomes", type=int, default=100, help="--config c5: number of genomes (a multiple of 5)")
    ap.add_argument("--batch", type=int, default=0, help="--config c5: jobs per rv_batch_run call (the library's own host threads, the anchor cascades' level loops of a "
                                                         "batch as one set of launches); 0 = a Python thread per job in flight (--jobs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and parity legs")
    ap.add_argument("--no-allcores", action="store_true", help="skip the all-host-cores CPU leg")
    ap.add_argument("--cpu-full", action="store_true", help="also time the CPU restatement on the WHOLE workload on this host (single thread: about 13 minutes and 17 GB "
                                                          "at 2 x 250 Mbp) and compare its anchors with the GPU's -- `cpu_baseline.full_size`")
    ap.add_argument("--no-check", action="store_true", help="skip the full-size property check of the last step's result")
    ap.add_argument("--prof-all", action="store_true", help="time every kernel class inside the timed region (adds events to every level)")
    ap.add_argument("--mode", choices=("auto", "per-rank", "divide"), default="auto",
                    help="N>1: 'divide' = ONE alignment divided over the ranks (frontier hand-off, reveal_amd/shard.py; strong scaling), "
                         "'per-rank' = one alignment per rank (weak); auto = divide for inputs of >= 100 Mbp")
    ap.add_argument("--divide", action="store_true", help="same as --mode divide")
    ap.add_argument("--jobs", type=int, default=1,
                    help="independent alignments per rank, run concurrently (one index handle, HIP stream and host thread each): "
                         "config 5's 20 independent 5-genome jobs on 8 GPUs (reveal/align.py:45-53); the default 1 is the metric's config")
    args = ap.parse_args()

    if args.class_one:
        return run_one_class(args)
    if args.classes:
        return run_classes(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ.get("WORLD_SIZE")))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if os.environ.get("RV_BENCH_DRYRUN"):      # launcher plumbing only (tests/test_cpu_host.py, no GPU): the ranks meet, count themselves, rank 0 reports
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_counted": int(t.item()), "gpus_flag": args.gpus}))
        if world > 1:
            dist.destroy_process_group()
        return
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("RV_BENCH_SHARE_GPU"):      # plumbing check on a one-GPU box: every rank on device 0, gloo instead of RCCL
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import numpy as np
    from reveal_amd import _lib, synth, check
    _lib.set_device(local_rank)
    if args.config == "c5":
        if args.L == 250_000_000:
            args.L = 5_000_000
        run_config5(args, rank, local_rank, world, dist, torch)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.config == "stream":
        run_stream(args, rank, local_rank, world, dist, torch)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    mode = "divide" if args.divide else args.mode
    big = args.L * args.genomes >= 100_000_000
    if world == 1:
        modes = ["per-rank"]
    elif mode == "auto":      # both in one invocation: throughput (weak) and the interval-split curve (strong)
        # (two samples: the anchor cascade finishes the run on rank 0 before there is a frontier to hand out -- rv_align_builtin_until returns 0 --,
        #  so the divided leg is not run: the line says "speedup 1.0, nothing to divide".  --mode divide still runs it.)
        modes = ["per-rank"] + (["divide"] if big and args.genomes > 2 else [])
    else:
        modes = [mode]
    jobs = max(1, args.jobs) if "per-rank" in modes else 1
    seqs = synth.family(args.L, args.genomes, seed=42 + (1000 * rank if "per-rank" in modes else 0), snp=args.snp, indelfrac=args.indelfrac, repeats=args.repeats, nruns=args.nruns)
    if args.snp != 0.01 or args.repeats or args.nruns:
        args.no_cpu = args.no_extra = True
    if args.contigs > 1:
        seqs = cut_into_contigs(seqs, args.contigs)
        args.no_cpu = args.no_extra = True
    bases = sum(len(s) for s in flat(seqs))
    idx = build_index(seqs, args.sa64)
    upload_ms = idx.upload_ms      # host->device copy of the assembled text (rv_upload), outside every timed region
    # further jobs of this rank: their own inputs (other seeds), handles and streams
    extra = [build_index(synth.genomes(args.L, args.genomes, seed=42 + 1000 * rank + 17 * j, indelfrac=args.indelfrac), args.sa64) for j in range(1, jobs)]
    # a divided run works on ONE input: rank 0's (seed 42); the other ranks hold its text
    idx_div = idx
    if "divide" in modes and "per-rank" in modes and rank != 0:
        idx_div = build_index(synth.genomes(args.L, args.genomes, seed=42, indelfrac=args.indelfrac), args.sa64)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    divide_state = {}

    def step(divide):
        if divide:      # rank 0 constructs and runs the top levels, the ranks pull sub-index batches from it
            # the hand-off over reveal_amd.transport: requests over local sockets, segments out of rank 0's HBM by HIP's inter-process copies -- no
            # tensor library, no collective on the data path (torch.distributed only brackets the timed region, as the contract says)
            from reveal_amd import shard, transport
            if "grp" not in divide_state:
                divide_state["grp"] = transport.Group.from_env()
            return shard.align_sharded_group(idx_div, divide_state["grp"], transport.DeviceMemory(idx_div._lib, local_rank), args.minl, args.minn)
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

    kname = "scan_multi" if args.genomes > 2 else "scan_pair"

    def timed(divide):
        """W untimed + K timed steps of one mode -> (max-over-ranks seconds, result of the last step, profile of the judged kernel)"""
        for _ in range(args.warmup):
            step(divide)
        # inside the timed region only the roofline-judged kernel is timed (HIP events riding on its own dispatch, on the library's stream);
        # the per-class breakdown comes from two extra, untimed steps with every class timed
        for ix in [idx, idx_div] + extra:
            ix.prof(enable=True, reset=True, only=None if args.prof_all else (kname,))
        barrier()
        t0 = time.perf_counter()
        last = None
        for _ in range(args.steps):
            last = None      # (the caller is done with a result before it asks for the next: its arrays are handed back to the library, rv_set_result_buffers)
            last = step(divide)
        barrier()
        elapsed = time.perf_counter() - t0
        if rank == 0 and not divide and not args.no_check and len(modes) > 1:
            last["T_after"] = idx.array("T")      # (the divided run of the same invocation works on the same handle afterwards)
        prof = (idx_div if divide else idx).prof(enable=False)
        for ix in extra:      # the judged kernel over all jobs (launches that overlap other jobs' kernels share the GPU with them)
            p2 = ix.prof(enable=False)
            prof = {k: tuple(a + b for a, b in zip(prof[k], p2[k])) for k in prof}
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if not os.environ.get("RV_BENCH_SHARE_GPU") else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, last, prof

    runs = {m: timed(m == "divide") for m in modes}
    primary = modes[0]
    divide = primary == "divide"
    tmax, last, prof = runs[primary]
    # full-size properties of the last timed step's result (reveal_amd/check.py), before the breakdown steps run again
    properties = full_size = None
    if rank == 0 and not args.no_check:
        T0 = np.frombuffer(b"$".join(flat(seqs)) + b"$", dtype=np.uint8)
        nsep = np.asarray(sample_seps(seqs), dtype=np.int64)
        if divide:      # (a divided run lower-cases each share on its own rank: rebuild the text from the merged anchors)
            from reveal_amd import shard
            T1 = shard.lower_text(T0, last["anchors"])
        else:
            T1 = last.pop("T_after", None)
            if T1 is None:
                T1 = idx.array("T")
        properties = check.recursion_properties(T0, T1, last["anchors"], nsep, args.minl, collinear=args.contigs <= 1)
        if divide:
            properties["text"] = "lower-cased text rebuilt from the merged anchors (each rank lower-cases its own share)"
        # the last timed step's anchor set and final text against the CPU path's digests at THIS size (tests/golden/fullsize.json:
        # the reference's divsufsort + the restated recursion, run once in the build container by oracle/gen_fullsize_golden.py)
        grec = check.golden_record(args.L, args.genomes, 42, args.indelfrac, args.minl, args.minn, snp=args.snp, repeats=args.repeats, nruns=args.nruns) if args.contigs <= 1 else None
        if grec is not None:
            full_size = check.compare_with_golden(grec, anchors=last["anchors"], T_final=T1)
            full_size["cpu_seconds_at_this_size"] = grec["cpu_seconds"]
        del T1
        if "divide" in runs and not divide:      # the divided run of the same invocation: its merged anchors against the same checks
            from reveal_amd import shard
            dl = runs["divide"][1]
            dp = check.recursion_properties(T0, shard.lower_text(T0, dl["anchors"]), dl["anchors"], nsep, args.minl)
            properties["divide_all"] = dp["all"]
            properties["divide_same_anchor_set_as_rank0_alone"] = anchor_set(*dl["anchors"]) == anchor_set(*last["anchors"])
        del T0
    breakdown = roofline_other = None
    if rank == 0 and not divide:
        idx.prof(enable=True, reset=True)
        for _ in range(2):
            idx.construct()
            idx.align_builtin(args.minl, args.minn)
        pall = idx.prof(enable=False)
        breakdown = {k: v[1] / 2 for k, v in pall.items() if v[0]}
        # the other kernels of the step against the same roofline (their own algorithmic bytes, include/reveal_amd.h RV_K_*), from these
        # two untimed steps: not the judged kernel, but what the step spends its time in
        roofline_other = {k: {"launches_per_step": v[0] / 2, "ms_per_step": v[1] / 2, "achieved_GBps": (v[2] / 1e9) / (v[1] / 1e3), "frac": (v[2] / 1e9) / (v[1] / 1e3) / HBM_PEAK_GBS}
                          for k, v in pall.items() if v[0] and v[1] > 0 and v[2] > 0 and k not in (kname, "sa_build", "cascade")}
    barrier()

    total_bases = float(bases * jobs) * (1 if divide else world)

    if rank == 0:
        launches, ms, nbytes = prof[kname]
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        traffic = traffic_source = None
        for name in ("pmc_scan_%dx%d-%d.json" % (args.genomes, args.L, 64 if args.sa64 else 32), "pmc_scan.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    # per-launch HBM bytes from the PMC passes only describe the workload they were collected on
                    if pj.get("workload") == "%dx%d-%d" % (args.genomes, args.L, 64 if args.sa64 else 32):
                        traffic = pj.get("hbm_bytes_per_launch")
                        # not measured in this run: rocprofv3's counter passes cannot run inside a timed bench
                        traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` on this workload, %s launches, file dated %s)" % (
                            name, pj.get("launches"), time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(pmc))))
                        break
                except Exception:
                    pass
        try:
            read_gbs, copy_gbs = _lib.measure_bandwidth(1 << 30, 10)      # the node's practical ceiling (streaming kernels, 1 GiB)
        except Exception:
            read_gbs = copy_gbs = None
        st = last["stats"]
        if divide:
            sharding = "one alignment divided over %d ranks (frontier hand-off after the top levels, no collective), ranks per share %s" % (world, last.get("shares"))
        elif jobs == 1:
            sharding = "one alignment per rank, no exchange"
        else:
            sharding = "%d independent alignments per rank, concurrently (a handle, stream and host thread each), no exchange" % jobs
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
            "config": {"workload": "%dx synthetic %g Mbp genomes (uniform ACGT, %s, seed 42%s), rem -m %d -n %d, construct + full recursion, "
                                   "bench picker; text resident in HBM before the timed region (host->device copy of the text not timed)"
                                   % (args.genomes, args.L / 1e6, ("%g%% SNP" % (100 * args.snp) + (", %g%% of the base in interspersed repeats + tandem arrays, %d runs of N" % (100 * args.repeats, args.nruns) if args.repeats or args.nruns else "")) if not args.indelfrac else
                                      "1%% mutation events of which %g%% indels with zipf(1.7) lengths: the reference simulator's model" % (100 * args.indelfrac),
                                      ("" if divide else "+1000*rank") + ("" if args.contigs <= 1 else "; every genome cut into %d contigs, the later samples' in another order" % args.contigs),
                                      args.minl, args.minn),
                       "bases_per_gpu": bases * jobs if not divide else bases / world, "bases_per_step": total_bases, "index": "64-bit" if args.sa64 else "32-bit",
                       "jobs_per_gpu": jobs, "sharding": sharding},
            # the text's way into HBM, outside the timed region (SURVEY 8(d) asks for it as a sub-timing): the host->device copy of the
            # assembled text, once per input; value_incl_upload charges it to every step
            "upload_ms": upload_ms,
            "value_incl_upload": total_bases * args.steps / (tmax + args.steps * upload_ms / 1e3) / 1e6,
            "roofline": {"bound": "hbm", "kernel": "k_full_scan" if args.genomes > 2 else "k_scan_pair",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "launches": launches, "avg_us": (ms * 1e3 / launches) if launches else None,
                         "algorithmic_bytes_per_launch": (nbytes / launches) if launches else None,
                         # two readings of the same launches, side by side: `frac` prices a rank at SURVEY 8(d)'s 8 B (4 B suffix + 4 B LCP value);
                         # the kernel streams 5 B of it -- the LCP value and the BWT byte that carries the separator's side -- and gathers suffixes
                         # for the ranks that pass only.  frac_of_bytes_streamed is what the memory system actually delivers of its peak.
                         "frac_model_8B": achieved / HBM_PEAK_GBS,
                         "bytes_streamed_per_rank": 5,
                         "frac_of_bytes_streamed": achieved * 5.0 / 8.0 / HBM_PEAK_GBS,
                         "frac_of_traffic": (traffic / (ms * 1e-3 / launches) / 1e9 / HBM_PEAK_GBS) if (traffic and launches and ms > 0 and args.genomes <= 2) else None,
                         "copy_peak": {"read_GBps": read_gbs, "copy_GBps_read_plus_write": copy_gbs, "bytes": 1 << 30,
                                       "frac_of_read_peak": (achieved / read_gbs) if read_gbs else None}},
            "breakdown_ms_per_step": breakdown,
            "roofline_other": roofline_other,
            # the judged kernel is the scan (8 B per rank); since the anchor cascade it runs once per step, and the step's time is
            # in the suffix-array build -- the kernel class with the most time per step and its own fraction, so the record does not hide it
            "roofline_largest_class_by_time": (lambda k: {"class": k, **roofline_other[k]})(max(roofline_other, key=lambda k: roofline_other[k]["ms_per_step"]))
                                              if roofline_other else None,
            "recursion": {"anchors": st["splits"], "anchored_bp": st["anchored_bp"], "levels": st["levels"], "subindices": st["steps"],
                          "scanned_ranks": st["scanned_ranks"], "host_s": st["t_host"], "scan_s": st["t_scan"],
                          "split_s": st["t_split"], "bubble_s": st["t_bubble"]},
            # the anchor cascade (rv_cascade.hip): an untraced two-sample run is decided from the top-level match list wherever that is
            # provably the reference's result, the rest is rebuilt from its text and finished by the leaf kernel; done = False: the
            # level pipeline (scan / split / bubble_sort per level) did the run
            "cascade": idx.cascade_info() if not divide else None,
            "sa_build": idx.sa_stats(),
            "properties_full_size": properties,
        }
        if world > 1 and "divide" not in runs and not divide and big and args.genomes <= 2:
            out["divide"] = {"speedup_vs_rank0_alone": 1.0, "ran": False,
                             "note": "nothing to divide: the anchor cascade decides a two-sample run on the rank that built the index before there is a frontier of "
                                     "sub-indices to hand out, and the suffix-array build (three quarters of the step) is not distributed; inputs with more than "
                                     "two samples are divided (--mode divide forces the leg).  On a node the ranks take one alignment each (this line) or a queue "
                                     "of inputs each (`stream` below)"}
        if "divide" in runs and not divide:
            dt, dl, _ = runs["divide"]
            out["divide"] = {"value": float(bases) * args.steps / dt / 1e6, "unit": "Mbp/s", "ms_per_step": dt / args.steps * 1e3, "scaling": "strong",
                             "what": "ONE alignment (rank 0's input) divided over the %d ranks in the same invocation: rank 0 constructs and runs the top "
                                     "levels, the ranks pull batches of the frontier's sub-indices from it, no collective" % world,
                             "ranks_per_share": dl.get("shares"), "batches_per_rank": dl.get("batches"),
                             "speedup_vs_rank0_alone": (tmax / args.steps) / (dt / args.steps),
                             "note": None if any(dl.get("shares") or [0]) else "nothing was handed out: the anchor cascade finished this two-sample run on rank 0 "
                                     "before there was a frontier to divide (rv_align_builtin_until); inputs with more than two samples are divided"}
        out["parity"] = {"full_size": full_size if full_size is not None else
                         "no CPU digests for this configuration in tests/golden/fullsize.json (oracle/gen_fullsize_golden.py writes them)"}
        if world == 1 and not args.no_extra and not divide and jobs == 1:
            # ---- companion figures, outside the timed region (what the headline does not show) ----
            # (1) the same workload through the LEVEL PIPELINE: scan / pick / label + split / bubble_sort of every level, reveal.c:731-1338 step by
            # step -- what index.align(mumpicker, graphalign) callers, rc = 1 and traced runs get; the headline's two-sample recursion
            # is the anchor cascade, which runs none of A10-A12
            idx.set_option("RV_NO_CASCADE", 1)
            try:
                idx.construct(); idx.align_builtin(args.minl, args.minn)
                idx.prof(enable=True, reset=True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                lp = None
                for _ in range(2):
                    lp = None
                    idx.construct()
                    lp = idx.align_builtin(args.minl, args.minn)
                torch.cuda.synchronize(); lp_s = (time.perf_counter() - t0) / 2
                pl = idx.prof(enable=False)
                lp_T = idx.array("T") if full_size is not None else None
                out["level_pipeline"] = {
                    "value": bases / lp_s / 1e6, "unit": "Mbp/s", "ms_per_step": lp_s * 1e3, "levels": lp["stats"]["levels"],
                    "kernel_classes_ms_per_step": {k: v[1] / 2 for k, v in pl.items() if v[0] and v[1] / 2 >= 0.05},
                    "roofline_by_class": {k: {"achieved_GBps": (v[2] / 1e9) / (v[1] / 1e3), "frac": (v[2] / 1e9) / (v[1] / 1e3) / HBM_PEAK_GBS}
                                          for k, v in pl.items() if v[0] and v[1] > 0 and v[2] > 0 and k in ("split", "bubble", "scan_pair", "scan_multi")},
                    "same_anchor_set_as_headline": check.anchor_digest(*lp["anchors"]) == check.anchor_digest(*last["anchors"]),
                    "golden": check.compare_with_golden(grec, anchors=lp["anchors"], T_final=lp_T) if full_size is not None else None,
                    "what": "RV_NO_CASCADE: construct + every level's scan / split / bubble_sort (2 steps after 1 warm-up, outside the timed region)"}
            finally:
                idx.set_option("RV_NO_CASCADE", 0)
            # (2) the same sizes with the reference simulator's own mutation model (indels: the second sample leaves the first's diagonal)
            if not args.indelfrac:
                iseqs = synth.genomes(args.L, args.genomes, seed=42, indelfrac=0.2)
                ibases = sum(len(x) for x in iseqs)
                iidx = build_index(iseqs, args.sa64)
                iidx.construct(); iidx.align_builtin(args.minl, args.minn)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ir = None
                for _ in range(2):
                    ir = None
                    iidx.construct()
                    ir = iidx.align_builtin(args.minl, args.minn)
                torch.cuda.synchronize(); i_s = (time.perf_counter() - t0) / 2
                irec = check.golden_record(args.L, args.genomes, 42, 0.2, args.minl, args.minn)
                iT0 = np.frombuffer(b"$".join(iseqs) + b"$", dtype=np.uint8)
                insep = np.cumsum([len(x) + 1 for x in iseqs])[:-1] - 1
                iT1 = iidx.array("T")
                out["indel"] = {
                    "value": ibases / i_s / 1e6, "unit": "Mbp/s", "ms_per_step": i_s * 1e3, "bases": ibases,
                    "workload": "%dx %g Mbp, 1%% mutation events of which 20%% indels (half insertions, half deletions, zipf(1.7) lengths <= 2000): "
                                "reveal_amd/synth.py after utils/simulate.py:17-77, seed 42" % (args.genomes, args.L / 1e6),
                    "anchors": ir["stats"]["splits"], "anchored_bp": ir["stats"]["anchored_bp"], "cascade": iidx.cascade_info(), "sa_build": iidx.sa_stats(),
                    "properties": check.recursion_properties(iT0, iT1, ir["anchors"], insep, args.minl)["all"],
                    "golden": check.compare_with_golden(irec, anchors=ir["anchors"], T_final=iT1) if irec is not None else None}
                del iidx, iseqs, iT0, iT1
        if world == 1 and not args.no_extra and not divide and jobs == 1 and args.L * args.genomes >= 100_000_000:
            # (3) north_star's target volume as a sustained stream: 20 inputs of this size (10 Gbp at the default workload) through `--config stream` in a
            # process of its own -- host assembly, host->device copy, construct, recursion and result delivery of every input inside its timed region
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--config", "stream", "--pairs", "20", "--steps", "1", "--warmup", "1", "--no-check",
                       "--L", str(args.L), "--genomes", str(args.genomes), "--minl", str(args.minl), "--minn", str(args.minn)] + (["--sa64"] if args.sa64 else [])
                pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
                sj = json.loads([x for x in pr.stdout.decode().splitlines() if x.startswith("{")][-1])
                out["stream"] = {"value": sj["value"], "unit": "Mbp/s", "includes_upload": True, "inputs": sj["config"]["inputs_per_rank_and_step"], "Gbp": sj["Gbp_total"],
                                 "wall_seconds": sj["wall_seconds"], "ms_per_input": sj["ms_per_input"], "in_flight_per_gpu": sj["config"]["in_flight_per_gpu"],
                                 "host_thread_ms_per_input": sj["host_thread_ms_per_input"],
                                 "what": "python bench.py --config stream --pairs 20: every input assembled from the caller's sequences, copied to HBM, constructed and "
                                         "anchored inside the timed region, four inputs in flight (a handle, HIP stream and host thread each)"}
            except Exception as e:      # noqa: BLE001  (a companion figure: its failure must not lose the line)
                out["stream"] = {"failed": repr(e)[:300]}
        if world == 1 and not args.no_cpu:
            # CPU legs and bit-exact parity on a stated sample: 2 x 20 Mbp (or the workload itself when it is not larger)
            cl = min(args.L, CPU_SAMPLE_L)
            same = cl == args.L
            cseqs = seqs if same else synth.genomes(cl, args.genomes, seed=42)
            cb = cpu_baseline(cseqs, args.minl, args.minn)
            cbases = sum(len(s) for s in cseqs)
            sample = "%dx %g Mbp (same generator, seed 42)" % (args.genomes, cl / 1e6)
            out["cpu_baseline"] = {
                "value": cbases / (cb["t_construct"] + cb["t_align"]) / 1e6, "unit": "Mbp/s", "cores": 1, "kind": "port",
                "sample": "%s, construct %.2f s + recursion %.2f s, single thread, SA by %s" % (
                    sample, cb["t_construct"], cb["t_align"],
                    "the reference's divsufsort (oracle/_ref)" if cb["ref_divsufsort"] else "the oracle's own sorter"),
                "host_cores_available": os.cpu_count(),
                "note": "measured on the sample, not on the workload: suffix sorting is n log n, so the CPU's rate per base at the full "
                        "workload size is lower than this figure -- the GPU/CPU ratio read off this line is on the generous-to-CPU side",
            }
            if args.cpu_full and not same:
                fb = cpu_baseline(seqs, args.minl, args.minn)
                frl, frn, froff, frpos = fb["result"]["anchors"]
                out["cpu_baseline"]["full_size"] = {
                    "value": bases / (fb["t_construct"] + fb["t_align"]) / 1e6, "unit": "Mbp/s", "cores": 1,
                    "sample": "the workload itself on this host: construct %.1f s + recursion %.1f s, single thread" % (fb["t_construct"], fb["t_align"]),
                    "identical_anchor_set": check.anchor_digest(frl, froff, frpos) == check.anchor_digest(*last["anchors"])}
                del fb
            if not args.no_allcores:
                ac = cpu_all_cores(cl, args.genomes, args.minl, args.minn)
                if ac:
                    out["cpu_baseline"]["all_cores"] = {
                        "value": ac["value"], "unit": "Mbp/s", "cores": ac["workers"],
                        "sample": "%d independent single-threaded alignments of %s at once (one process per core this process may use -- "
                                  "%d by affinity and CPU quota -- capped at 64 and by free memory), %.1f s from first start to last end"
                                  % (ac["workers"], sample.replace("seed 42", "seeds 5000+13k"), ac["cores"], ac["seconds"])}
            # the same sample through the HIP path: anchors and final text must be identical to the CPU path's
            if same:
                gres, gT = last, idx.array("T").tobytes()
            else:
                pidx = build_index(cseqs, args.sa64)
                pidx.construct()
                gres = pidx.align_builtin(args.minl, args.minn)
                gT = pidx.array("T").tobytes()
            rl, rn, roff, rpos = cb["result"]["anchors"]
            ra = anchor_set(rl, roff, rpos)
            ga = anchor_set(*gres["anchors"])
            out["parity"].update({"sample": sample, "anchors_gpu": len(ga), "anchors_cpu": len(ra), "identical_anchor_set": ra == ga,
                                  "identical_final_text": gT == cb["result"]["T"]})
    else:
        out = None
    if world > 1 and big and not args.no_extra and not divide and jobs == 1:
        # north_star's target on a node: every rank its own queue of 20 inputs of this size (10 Gbp per rank at the default workload), host assembly and
        # host->device copies inside the timed region -- `--config stream` run by all ranks right here (its barriers and max-over-ranks timing are its own)
        import copy
        sa = copy.copy(args)
        sa.pairs, sa.steps, sa.warmup, sa.no_check = 20, 1, 1, True
        del idx, extra
        try:
            sj = run_stream(sa, rank, local_rank, world, dist, torch, emit=False)
            if rank == 0 and sj is not None:
                out["stream"] = {"value": sj["value"], "unit": "Mbp/s", "includes_upload": True, "n_gpus": world, "inputs_per_rank": sj["config"]["inputs_per_rank_and_step"],
                                 "Gbp": sj["Gbp_total"], "wall_seconds": sj["wall_seconds"], "ms_per_input": sj["ms_per_input"], "in_flight_per_gpu": sj["config"]["in_flight_per_gpu"],
                                 "host_thread_ms_per_input": sj["host_thread_ms_per_input"],
                                 "what": "every rank a queue of 20 inputs (`--config stream --pairs 20`): assembled from the caller's sequences, copied to HBM, constructed and "
                                         "anchored inside the timed region, four in flight per GPU; the time is the slowest rank's, barrier to barrier"}
        except Exception as e:      # noqa: BLE001  (a companion figure: its failure must not lose the line)
            if rank == 0:
                out["stream"] = {"failed": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
