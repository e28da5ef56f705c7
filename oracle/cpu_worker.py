#!/usr/bin/env python3
"""One CPU-baseline job of bench.py's `cpu_baseline` leg -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

    python oracle/cpu_worker.py --L 5000000 --genomes 2 --seed 42 --minl 20 --minn 2 [--start-at <unix time>]

Generates its inputs (reveal_amd/synth.py: pure numpy), optionally sleeps until a common
start time (so that P workers launched together load the host's cores at the same moment:
the "all host cores" variant of SURVEY.md 8(d) = P independent single-threaded alignments,
the only parallelism the reference has, reveal/align.py:45-53), runs construct + the full
recursion with the benchmark callbacks on the CPU restatement (oracle/reveal_oracle.c, SA by
the reference's own divsufsort when oracle/_ref is present) and prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, required=True)
    ap.add_argument("--genomes", type=int, default=2)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--minl", type=int, default=20)
    ap.add_argument("--minn", type=int, default=2)
    ap.add_argument("--start-at", type=float, default=0.0)
    args = ap.parse_args()
    from reveal_amd import synth               # pure numpy
    from oracle import oracle_ctypes
    seqs = synth.genomes(args.L, args.genomes, seed=args.seed)
    O = oracle_ctypes.Oracle(False)
    T, nsep, nodes = bytearray(), [], []
    for k, s in enumerate(seqs):
        if k:
            nsep.append(len(T) - 1)
        b = len(T)
        T += s + b"$"
        nodes.append((b, len(T) - 1))
    T = bytes(T)
    late = 0.0
    if args.start_at > 0:
        d = args.start_at - time.time()
        if d > 0:
            time.sleep(d)
        else:
            late = -d
    t0 = time.perf_counter()
    c = O.construct(T, nsep, len(seqs))
    t1 = time.perf_counter()
    r = O.align_bench(c, nodes, args.minl, args.minn)
    t2 = time.perf_counter()
    print(json.dumps(dict(bases=sum(len(s) for s in seqs), t_construct=t1 - t0, t_align=t2 - t1, anchors=int(len(r["anchors"][0])),
                          late=late, ref_divsufsort=bool(O.ref_divsufsort), t_begin=time.time() - (t2 - t0), t_end=time.time())))


if __name__ == "__main__":
    main()
