"""Deterministic synthetic inputs for the golden vectors and the parity tests.

Pure integer hashing (splitmix64 finaliser) so the same (seed, n, kind) gives
the same bytes on every numpy version and every machine.  Used by
tests/golden/make_golden.py (which records what the compiled reference returns
for these inputs) and by the tests (which regenerate inputs for the large
cases whose fixtures hold only digests).
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash_u64(seed: int, n: int, first: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = np.arange(first, first + n, dtype=np.uint64) + np.uint64(seed) * _GOLD
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return x


F32_SPECIALS = np.array(
    [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754942e-38, 1.17549435e-38,
     3.4028235e38, -3.4028235e38, 0.5, 2.0, 1.0000001, 0.99999994, 16777216.0, 16777217.0, 1e-20, 1e20],
    dtype=np.float32)
F64_SPECIALS = np.array(
    [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308,
     1.7976931348623157e308, -1.7976931348623157e308, 0.5, 2.0, 1e-200, 1e200],
    dtype=np.float64)
I32_SPECIALS = np.array([0, 1, -1, 2, -2, 2147483647, -2147483648, 2147483646, -2147483647, 65536, -65536,
                         46341, -46341, 3, 7], dtype=np.int32)


def gen(dtype, n: int, seed: int, kind: str = "mixed") -> np.ndarray:
    """n values of `dtype`.

    kind: "uniform"  floats in [-4, 4) / ints in [-1000, 1000]
          "wide"     random bit patterns (floats: any exponent, incl. NaN/Inf/denormal;
                     ints: full range)
          "mixed"    specials first (as many as fit), then alternating uniform / wide
          "nonzero"  like "mixed" for ints but never 0 and never the INT_MIN/-1 pair
                     partner (-1): divisors the reference can divide by without trapping
          "positive" floats in (0.01, 100): BASELINE config 4's base distribution
    """
    dtype = np.dtype(dtype)
    h = hash_u64(seed, n)
    if dtype == np.float32:
        uni = ((h >> np.uint64(40)).astype(np.float64) * 2.0 ** -24 * 8.0 - 4.0).astype(np.float32)
        wide = (h & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
        spec = F32_SPECIALS
    elif dtype == np.float64:
        uni = (h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 * 8.0 - 4.0
        wide = h.view(np.float64)
        spec = F64_SPECIALS
    elif dtype == np.int32:
        uni = ((h >> np.uint64(33)) % np.uint64(2001)).astype(np.int64).astype(np.int32) - np.int32(1000)
        wide = (h & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
        spec = I32_SPECIALS
    elif dtype == np.int64:
        uni = ((h >> np.uint64(33)) % np.uint64(2001)).astype(np.int64) - np.int64(1000)
        wide = h.view(np.int64)
        spec = I32_SPECIALS.astype(np.int64)
    else:
        raise TypeError(dtype)
    if kind == "uniform":
        return uni.copy()
    if kind == "wide":
        return wide.copy()
    if kind == "positive":
        assert dtype.kind == "f"
        u = (h >> np.uint64(40)).astype(np.float64) * 2.0 ** -24
        return (0.01 + u * 99.99).astype(dtype)
    out = np.where((np.arange(n) & 1) == 0, uni, wide).astype(dtype)
    k = min(len(spec), n)
    out[:k] = spec[:k]
    if kind == "nonzero":
        assert dtype.kind == "i"
        out[out == 0] = 3
        out[out == -1] = 5
    elif kind != "mixed":
        raise ValueError(kind)
    return out
