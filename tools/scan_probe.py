"""time the pair MUM scan kernel alone (top-level getmums) at several index sizes"""
import sys, time
sys.path.insert(0, ".")
from reveal_amd import reveallib, synth
for L in [int(x) for x in sys.argv[1:]] or [5_000_000]:
    seqs = synth.genomes(L, 2)
    idx = reveallib.index()
    for k, s in enumerate(seqs):
        idx.addsample("g%d" % k); idx.addsequence(s)
    t0 = time.perf_counter(); idx.construct(); t1 = time.perf_counter()
    idx.getmums(20)
    idx.prof(enable=True, reset=True)
    for _ in range(10):
        nm = len(idx.getmums(20))
    p = idx.prof(enable=False)["scan_pair"]
    print("n=%d construct %.1f ms | scan: %d launches, %.1f us avg, %.0f GB/s algorithmic (8 B/rank), %d mums | sa %s" % (
        idx.n, (t1 - t0) * 1e3, p[0], p[1] * 1e3 / p[0], p[2] / p[1] / 1e6, nm, idx.sa_stats()))
    del idx
