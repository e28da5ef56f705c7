"""time the top-level scan kernels alone at several index sizes: getmums (two samples), or with --multi K getmultimums(20, 2) of K genomes
usage: python tools/scan_probe.py [--multi K] L [L ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reveal_amd import reveallib, synth
argv = sys.argv[1:]
K = 2
if argv and argv[0] == "--multi":
    K = int(argv[1]); argv = argv[2:]
for L in [int(x) for x in argv] or [5_000_000]:
    seqs = synth.genomes(L, K)
    idx = reveallib.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k); idx.addsequence(s)
    t0 = time.perf_counter(); idx.construct(); t1 = time.perf_counter()
    scan = (lambda: idx.getmums(20)) if K == 2 else (lambda: idx.getmultimums(20, 2))
    cls = "scan_pair" if K == 2 else "scan_multi"
    scan()
    idx.prof(enable=True, reset=True)
    t2 = time.perf_counter()
    for _ in range(3 if K > 2 else 10):
        nm = len(scan())
    t3 = time.perf_counter()
    p = idx.prof(enable=False)[cls]
    print("n=%d construct %.1f ms | %s: %d launches, %.1f us avg, %.0f GB/s algorithmic (8 B/rank), %d matches, %.1f ms per call incl. lists | sa %s" % (
        idx.n, (t1 - t0) * 1e3, cls, p[0], p[1] * 1e3 / p[0], p[2] / p[1] / 1e6, nm, (t3 - t2) * 1e3 / (3 if K > 2 else 10), idx.sa_stats()))
    del idx
