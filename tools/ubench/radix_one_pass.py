"""One radix pass on 8 key bits through the 256-bin kernel (flags 6) and the 1024-bin kernel (flags 7: what a symbol-aligned MSD digit of 625 bins
would run on), uniform and DNA-key digits: what an MSD hybrid's global passes would cost.  python tools/ubench/radix_one_pass.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reveal_amd import _lib
lib = _lib.get(False)
n = 290_000_000
for flags in (6, 7):
    for dist in (0, 1):
        ms = (ctypes.c_double * 4)(); bad = ctypes.c_int64(-1)
        assert lib.dll.rv_test_radix_time(n, 8, dist, flags | 8, 4, ms, ctypes.byref(bad)) == 0
        print("flags", flags, "dist", dist, "one pass on 8 bits: %.2f ms" % min(ms[1:]), "bad", bad.value)
