import os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz
from helpers import assemble, feed, oracle
from reveal_amd import reveallib, reveallib64
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for ncase in range(want + 1):
    seqs, minl = fuzz.make_case(rng)
    seqs = fuzz.maybe_contigs(rng, seqs)
    sa64 = rng.random() < 0.2
    if ncase != want:
        for _ in range(2):
            rng.choice([2, 3, 8, 32]); rng.choice([2, 3, 5])
T, nsep, nodes = assemble(seqs)
O = oracle(sa64)
c = O.construct(T, nsep, len(seqs))
print("case", want, "sa64", sa64, "n", len(T), "samples", len(seqs), "maxlcp", int(c["LCP"].max()), [len(x) if isinstance(x, str) else [len(y) for y in x] for x in seqs])
for env in [{}, {"RV_NO_TEXT_JUMP": "1"}, {"RV_NO_LCP_LIST": "1"}, {"RV_NO_LCP_LIST": "1", "RV_NO_TEXT_JUMP": "1"}, {"RV_NO_FAR_TWINS": "1", "RV_NO_LCP_LIST": "1", "RV_NO_TEXT_JUMP": "1"},
            {"RV_NO_FAR_TWINS": "1", "RV_NO_TEXT_JUMP": "1"}, {"RV_NO_FAR_TWINS": "1", "RV_NO_LCP_LIST": "1"}]:
    for k in list(os.environ):
        if k.startswith("RV_"):
            del os.environ[k]
    os.environ.update(env)
    idx = feed((reveallib64 if sa64 else reveallib).index(), seqs)
    idx.construct()
    sa, lcp = idx.array("SA"), idx.array("LCP")
    bad = np.nonzero(sa != c["SA"])[0]; badl = np.nonzero(lcp != c["LCP"])[0]
    print(env, idx.sa_stats(), "SA bad", len(bad), bad[:6], [(int(sa[k]), int(c["SA"][k]), int(c["LCP"][k])) for k in bad[:4]], "LCP bad", len(badl), badl[:6])
