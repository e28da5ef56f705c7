"""device primitives (scan, radix sort) against numpy -- through the C ABI test hooks"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from reveal_amd import _lib
    L = _lib.get(False)
    assert L.dll.rv_device_count() > 0, "no HIP device"
    return L


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 100000, 2048 * 2048 + 17])
def test_exclusive_sum(lib, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 5, size=n, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    assert lib.dll.rv_test_exclusive_sum_u32(a.ctypes.data, out.ctypes.data, n) == 0, lib.err()
    ref = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]).astype(np.uint32)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("n", [1, 7, 64, 2049, 300000])
def test_inclusive_max(lib, n):
    rng = np.random.default_rng(n)
    a = np.where(rng.random(n) < 0.05, np.arange(n), 0).astype(np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    assert lib.dll.rv_test_inclusive_max_u32(a.ctypes.data, out.ctypes.data, n) == 0, lib.err()
    assert np.array_equal(out, np.maximum.accumulate(a))


@pytest.mark.parametrize("n,lo,hi", [(1, 0, 64), (2, 0, 8), (1000, 0, 64), (4096, 0, 16), (4097, 8, 40),
                                      (123457, 0, 64), (1 << 20, 0, 24), (3000001, 32, 56), (70001, 3, 40), (5000, 0, 5)])
def test_radix_sort_stable(lib, n, lo, hi):
    rng = np.random.default_rng(n + lo)
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    if n > 1000:                       # many duplicates in the sorted bit range: stability matters
        keys[: n // 2] &= np.uint64(0xFFFF00FF00FF00FF)
    vals = np.arange(n, dtype=np.uint32)
    k, v = keys.copy(), vals.copy()
    assert lib.dll.rv_test_radix_sort(k.ctypes.data, v.ctypes.data, n, lo, hi) == 0, lib.err()
    width = hi - lo
    mask = np.uint64((1 << width) - 1) if width < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    # (the last pass masks its digit: exactly the bits [lo, hi) take part)
    sub = (keys >> np.uint64(lo)) & mask
    order = np.argsort(sub, kind="stable")
    assert np.array_equal(v, vals[order])
    assert np.array_equal(k, keys[order])


@pytest.mark.parametrize("flags", range(16))
@pytest.mark.parametrize("n,bits,dist", [(100_000, 40, 1), (4096 * 9 + 5, 37, 0), (300_000, 16, 2), (1_200_000, 40, 1), (1_050_001, 35, 0)])
def test_radix_sort_variants(lib, flags, n, bits, dist):
    """every variant of the sort (10-bit digits, XCD-aware tile order, 16-bit wave counters, flag 8: histograms from the keys instead of the
    digit bytes the pass before left -- those only above 2^20 keys: rv_prims.hip) sorts device-made
    keys stably -- checked on the device: no adjacent pair out of order by (key bits, original index)"""
    import ctypes
    ms = (ctypes.c_double * 2)()
    bad = ctypes.c_int64(-1)
    assert lib.dll.rv_test_radix_time(n, bits, dist, flags, 2, ms, ctypes.byref(bad)) == 0, lib.err()
    assert bad.value == 0
