import ctypes, os, sys
sys.path.insert(0, "/root/repo")
from reveal_amd import _lib
lib = _lib.get(False)
n = 290_000_000
for flags in (6, 7):
    for dist in (0, 1):
        ms = (ctypes.c_double * 4)(); bad = ctypes.c_int64(-1)
        assert lib.dll.rv_test_radix_time(n, 8, dist, flags | 8, 4, ms, ctypes.byref(bad)) == 0
        print("flags", flags, "dist", dist, "one pass on 8 bits: %.2f ms" % min(ms[1:]), "bad", bad.value)
