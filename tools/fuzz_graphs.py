#!/usr/bin/env python3
"""Randomised parity soak for GRAPH inputs: `reveal rem` on graphs (and graphs + FASTA, multi-sequence samples) with readers, picker and graphalign inside the library
(rv_gfa_parse / rv_graph_adopt, rv_set_graph_picker) against the same job through the Python readers and callbacks -- the two GFA files must be equal byte for byte, and every
input must be spelled by its path.  Random families (SNPs, indels, repeats, runs of N), random partitions into input graphs, random picker options.  Test infrastructure.
usage: python tools/fuzz_graphs.py [seconds] [seed]      (FUZZ_BIG=1: larger families)"""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import graphrem_cases as C  # noqa: E402
from reveal_amd import rem, schemes, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed0)
t0 = time.time()
cases = jobs = 0
while time.time() - t0 < budget:
    cases += 1
    seed = rng.randrange(1 << 30)
    big = bool(os.environ.get("FUZZ_BIG"))      # FUZZ_BIG=1: up to 14 genomes of up to 400 kbp
    K = rng.randint(4, 14 if big else 9)
    L = rng.choice([50000, 150000, 400000] if big else [3000, 8000, 20000, 50000])
    fam = dict(snp=rng.choice([0.002, 0.01, 0.03, 0.08]), indelfrac=rng.choice([0.0, 0.0, 0.2, 0.5]), repeats=rng.choice([0.0, 0.0, 0.03, 0.1]), nruns=rng.choice([0, 0, 2]))
    args = dict(trim=rng.random() < 0.8, seedsize=rng.choice([10000, 10000, 300, 50]), maxmums=rng.choice([1000, 1000, 50, 10]), gcmodel=rng.choice(["sumofpairs", "sumofpairs", "star-avg", "star-med"]),
                wpen=rng.choice([1, 1, 3]), wscore=rng.choice([1, 1, 2]))
    minl = rng.choice([20, 20, 12, 30])
    d = tempfile.mkdtemp(prefix="fuzzg_")
    seqs = synth.family(L, K, seed=seed, **fam)
    files = []
    for k, s in enumerate(seqs):
        p = os.path.join(d, "s%d.fa" % k)
        open(p, "w").write(">m%d\n%s\n" % (k, s.decode()))
        files.append(p)
    # a random partition into 2-4 groups; groups of one stay FASTA, the others become graphs first
    order = list(range(K)); rng.shuffle(order)
    ng = rng.randint(2, min(4, K))
    cuts = sorted(rng.sample(range(1, K), ng - 1))
    groups = [order[a:b] for a, b in zip([0] + cuts, cuts + [K])]
    inputs = []
    for gi, g in enumerate(groups):
        if len(g) == 1:
            inputs.append(files[g[0]])
        else:
            inputs.append(rem.graph_rem([files[k] for k in g], os.path.join(d, "g%d.gfa" % gi), minlength=minl)[2])
    texts = {}
    for native in (False, True):
        G, idx, fn = rem.graph_rem(inputs, os.path.join(d, "out%d.gfa" % native), args=schemes.PickerArgs(**args), native=native, preselect=False, minlength=minl)
        texts[native] = open(fn).read()
        jobs += 1
    ok = texts[True] == texts[False]
    if ok:
        spelled, _ = C.spelled_by_file(os.path.join(d, "out1.gfa"))
        ok = spelled == {"m%d" % k: s.decode().upper() for k, s in enumerate(seqs)}
    if not ok:
        print("MISMATCH seed0=%d case=%d seed=%d K=%d L=%d fam=%s args=%s minl=%d groups=%s dir=%s" % (seed0, cases, seed, K, L, fam, args, minl, groups, d))
        sys.exit(1)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
print("fuzz_graphs: %d cases (%d jobs both ways) in %.0f s, all equal" % (cases, jobs, time.time() - t0))
