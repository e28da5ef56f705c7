#!/usr/bin/env python3
"""The library's radix sort on device-made keys, every variant in one process (rv_test_radix_time: HIP events around the whole sort).
usage: python tools/ubench/radix_time.py [n] [bits]      default: 2.9e8 pairs (what 2 x 250 Mbp sorts), 40 bits"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reveal_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 290_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = _lib.get(False)
IT = 4
for dist, dname in ((1, "dna17"), (0, "uniform"), (2, "constant")):
    for flags in (0, 6, 6 | 8, 7):      # flag 8: no digit bytes (the histogram passes read the keys)
        ms = (ctypes.c_double * IT)()
        bad = ctypes.c_int64(-1)
        r = lib.dll.rv_test_radix_time(n, bits, dist, flags, IT, ms, ctypes.byref(bad))
        assert r == 0, lib.err()
        passes = (bits + (9 if flags & 1 else 7)) // (10 if flags & 1 else 8)
        best = min(ms[1:])
        print("%-8s digits %2d xcd %d cnt16 %d digit-bytes %d  passes %d  %.2f ms (%.2f per pass, %.2f TB/s on 24 B per pair and pass + 8 B histogram read)  bad %d" % (
            dname, 10 if flags & 1 else 8, (flags >> 1) & 1, (flags >> 2) & 1, 0 if flags & 8 or flags & 1 else 1, passes, best, best / passes, passes * 32.0 * n / best / 1e9, bad.value), flush=True)
