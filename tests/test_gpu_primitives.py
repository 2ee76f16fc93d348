"""The library's own device-wide primitives (csrc/rb_sort.hip: scans, csrc/rb_group.hip: LSD radix sorts out of the
grouping stage's stable partition passes) against numpy.  They replaced rocPRIM in round 4; every scan / sort of the
insert pipeline, the conflict path, the FASTQ / FASTA record finders and the sketch sets goes through them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scan(a, misalign=0):
    from rnabloom import _native as N
    a = np.ascontiguousarray(a, np.uint32)
    out = np.empty_like(a)
    N.check(N.lib.rb_debug_scan_u32(0, a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p), misalign))
    return out


def _excl(a):
    c = np.cumsum(a.astype(np.uint64)) & 0xFFFFFFFF
    return np.concatenate([[0], c[:-1]]).astype(np.uint32) if a.size else a.astype(np.uint32)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1023, 1024, 4095, 4096, 4097, 32767, 32768, 32769, 49153, 16384 * 3, 16384 * 3 + 1, 1_000_003, 23_456_789])
def test_exclusive_scan_matches_numpy(n):
    rng = np.random.default_rng(n + 7)
    a = rng.integers(0, 33, n, dtype=np.uint32)
    for mis in (0, 1, 3, 4, 13):          # input / output offsets from a 16-byte boundary: (0,0) aligned = the vectorised path, (1,0), (3,0), (0,1), (1,3)
        assert np.array_equal(_scan(a, mis), _excl(a)), (n, mis)


def test_exclusive_scan_wraps_like_u32_and_handles_large_values():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2**32, 100_001, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(_scan(a), _excl(a))
    ones = np.ones(5_000_000, np.uint32)
    assert np.array_equal(_scan(ones), np.arange(5_000_000, dtype=np.uint32))


def _sort(keys, vals, lo, hi, lo2=-1, hi2=-1, vals64=False):
    from rnabloom import _native as N
    k = np.ascontiguousarray(keys, np.uint64).copy()
    v = None if vals is None else np.ascontiguousarray(vals, np.uint64 if vals64 else np.uint32).copy()
    N.check(N.lib.rb_debug_sort_pairs(0, k.ctypes.data_as(C.c_void_p), None if v is None else v.ctypes.data_as(C.c_void_p), int(vals64), k.size, lo, hi, lo2, hi2))
    return k, v


def _ref_order(keys, ranges):
    """stable order on the concatenation of the bit ranges (higher range = more significant)"""
    sk = np.zeros(keys.size, np.uint64)
    shift = 0
    for lo, hi in ranges:
        if hi > lo:
            sk |= ((keys >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1)) << np.uint64(shift)
            shift += hi - lo
    return np.argsort(sk, kind="stable")


@pytest.mark.parametrize("n", [1, 2, 100, 4095, 4096, 4097, 70_001, 250_000, 1_300_000])
@pytest.mark.parametrize("bits", [(0, 64), (32, 57), (0, 1), (0, 10), (0, 11), (0, 20), (0, 21), (5, 5), (3, 40)])
def test_lsd_sort_pairs_is_a_stable_sort_on_the_bit_range(n, bits):
    rng = np.random.default_rng(n * 131 + bits[0] * 7 + bits[1])
    keys = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    if n > 1000:
        keys[rng.integers(0, n, n // 3)] = keys[rng.integers(0, n, n // 3)]      # duplicates: stability shows
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort(keys, vals, *bits)
    order = _ref_order(keys, [bits])
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])


@pytest.mark.parametrize("n", [1, 5000, 333_333])
def test_lsd_sort_keys_only_two_ranges_and_wide_values(n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**63, n, dtype=np.uint64)
    k, _ = _sort(keys, None, 0, 64)
    assert np.array_equal(k, np.sort(keys))
    # the conflict path's keys: (component label << 32) | occurrence id, occurrence ids below 2^occ_bits
    occ_bits, label_bits = 27, 21
    keys = (rng.integers(0, 2**label_bits, n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 2**occ_bits, n, dtype=np.uint64)
    vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    k, v = _sort(keys, vals, 0, occ_bits, 32, 32 + label_bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])
    v64 = rng.integers(0, 2**63, n, dtype=np.uint64)
    k, v = _sort(keys, v64, 0, 64, vals64=True)
    assert np.array_equal(k, keys[order]) and np.array_equal(v, v64[order])
