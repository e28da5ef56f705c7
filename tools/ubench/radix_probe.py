#!/usr/bin/env python3
"""One radix pass (rv_test_radix_sort, bits 0..8) over n keys whose digit follows different distributions; run under
rocprofv3 --kernel-trace and read the k_rs_scatter durations in launch order (tools/rocpd_stats.py prints totals only).
usage: python tools/ubench/radix_probe.py [log2 n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reveal_amd import _lib

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << lg
lib = _lib.get(False)
rng = np.random.default_rng(1)
vals = np.arange(n, dtype=np.uint32)
hi = rng.integers(0, 1 << 31, size=n, dtype=np.uint64) << np.uint64(8)
for name, digit in (("uniform256", rng.integers(0, 256, size=n, dtype=np.uint64)),
                    ("uniform16", rng.integers(0, 16, size=n, dtype=np.uint64)),
                    ("uniform4", rng.integers(0, 4, size=n, dtype=np.uint64)),
                    ("constant", np.zeros(n, dtype=np.uint64)),
                    ("sorted256", np.sort(rng.integers(0, 256, size=n, dtype=np.uint64)))):
    keys = (hi | digit).astype(np.uint64)
    v = vals.copy()
    r = lib.dll.rv_test_radix_sort(keys.ctypes.data, v.ctypes.data, n, 0, 8)
    assert r == 0, lib.err()
    assert np.all(np.diff((keys & np.uint64(255)).astype(np.int64)) >= 0)
    print(name, "ok", flush=True)
