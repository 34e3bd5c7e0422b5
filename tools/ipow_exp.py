import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
n = 1 << 28
rng = np.random.default_rng(1)
def t(fill_b):
    a = lib.full((n,), 3, np.int32); b = lib.empty((n,), np.int32); out = lib.empty((n,), np.int32)
    h = fill_b(1 << 20).astype(np.int32)
    for i in range(0, n, 1 << 20): lib.upload(b.ptr + i * 4, h)
    args = (C.c_int(4), C.c_int(2), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(out.ptr), C.c_size_t(n))
    for _ in range(20): lib.c.smhip_contiguous(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(50): lib.c.smhip_contiguous(*args)
    lib.record(e1); lib.synchronize()
    us = lib.elapsed_ms(e0, e1) / 50 * 1000
    return us, 12.0 * n / us * 1e-3 / 80
for name, f in (("exponent 3 everywhere", lambda k: np.full(k, 3)), ("exponents 0..7", lambda k: rng.integers(0, 8, k)), ("exponents 0..31", lambda k: rng.integers(0, 32, k)),
                ("exponents 0..2^20", lambda k: rng.integers(0, 1 << 20, k)), ("negative exponents", lambda k: -rng.integers(1, 100, k))):
    us, pct = t(f)
    print("i32 array ^ array, %-24s %8.1f us %5.1f%%" % (name, us, pct), flush=True)
