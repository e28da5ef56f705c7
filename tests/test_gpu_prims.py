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
                                      (123457, 0, 64), (1 << 20, 0, 24), (3000001, 32, 56)])
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
    # the sort sees 8-bit digits: bits [lo, lo + 8*ceil(width/8))
    width8 = min(64 - lo, 8 * ((width + 7) // 8))
    mask = np.uint64((1 << width8) - 1) if width8 < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    sub = (keys >> np.uint64(lo)) & mask
    order = np.argsort(sub, kind="stable")
    assert np.array_equal(v, vals[order])
    assert np.array_equal(k, keys[order])
