import os, sys, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import fuzz
from helpers import assemble, feed, oracle
from reveal_amd import reveallib
rng = random.Random(31)
for ncase in range(27):
    seqs, minl = fuzz.make_case(rng)
    seqs = fuzz.maybe_contigs(rng, seqs)
    sa64 = rng.random() < 0.2
    if ncase < 26:
        for _ in range(2):
            rng.choice([2, 3, 8, 32]); rng.choice([2, 3, 5])
print("case", ncase, "minl", minl, "contigs", [len(c) for c in seqs[0]], [len(c) for c in seqs[1]])
T, nsep, nodes = assemble(seqs)
O = oracle(False)
c = O.construct(T, nsep, 2)
ref = O.align_bench(c, nodes, minl, 2)
ra = fuzz.anchors_set(ref["anchors"])
for chain in (1, 0):
    idx = feed(reveallib.index(), seqs)
    idx.set_option("RV_NO_CASCADE_CHAIN", 0 if chain else 1)
    idx.set_option("RV_CASCADE_LOG", 1)
    idx.construct()
    got = idx.align_builtin(minl, 2)
    ga = fuzz.anchors_set(got["anchors"])
    print("chain", chain, "info", idx.cascade_info(), "anchors", len(ga), "ref", len(ra), "equal", ga == ra)
    if ga != ra:
        print(" only gpu:", sorted(set(ga) - set(ra))[:10]); print(" only ref:", sorted(set(ra) - set(ga))[:10])
print("nodes", nodes)
c = O.construct(T, nsep, 2)
ref = O.align_bench(c, nodes, minl, 2, trace_cap=5000)
tr = ref["trace"]
for t in tr:
    if t["nnodes"] > 2 or t["depth"] < 6 and t["nnodes"] >= 2 and t["n"] > 3000:
        print("ref visit depth", t["depth"], "n", t["n"], "nnodes", t["nnodes"], "nsamples", t["nsamples"], "nmums", t["nmums"], "picked", t["picked"], "l", t["l"], "sp_min", t["sp_min"])
