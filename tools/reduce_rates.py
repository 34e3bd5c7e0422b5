"""Whole-array reductions at N = 2^28 (2^27 for 8-byte types): sum, dot, fused add+sum, async forms (no host read-back)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
def timeit(fn, steps=50):
    for _ in range(5): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
print("%-34s %10s %9s %7s" % ("reduction", "us", "GB/s", "% peak"))
for dt in (np.float32, np.float64, np.int32, np.int64):
    n = (1 << 30) // np.dtype(dt).itemsize
    a = lib.empty((n,), dt); b = lib.empty((n,), dt); c = lib.empty((n,), dt)
    lib.c.smhip_fill(C.c_int(sma.DTYPES[np.dtype(dt)]), C.c_void_p(a.ptr), np.array([1], dtype=dt).ctypes.data_as(C.c_void_p), C.c_size_t(n))
    lib.c.smhip_fill(C.c_int(sma.DTYPES[np.dtype(dt)]), C.c_void_p(b.ptr), np.array([2], dtype=dt).ctypes.data_as(C.c_void_p), C.c_size_t(n))
    sp = lib.alloc(8)
    esz = np.dtype(dt).itemsize
    for name, fn, byts in (("sum", lambda: lib.sum_async(a, sp), esz * n), ("dot", lambda: lib.dot_async(a, b, sp), 2 * esz * n),
                           ("fused add+sum", lambda: lib.contiguous_sum_async(sma.OP_ADD, a, b, c, sp), 3 * esz * n)):
        t = timeit(fn)
        print("%-34s %10.1f %9.0f %6.1f%%" % ("%s %s n=2^%d" % (np.dtype(dt).name, name, n.bit_length() - 1), t, byts / t * 1e-3, byts / t * 1e-3 / 80), flush=True)
    lib.free(sp); del a, b, c
# complex<double> dot: the kernels alone (async entry point), then the synchronous call with its 16-byte read-back
import time
n = 1 << 26
a = lib.empty((2 * n,), np.float64); b = lib.empty((2 * n,), np.float64)
one = np.array([1.0], dtype=np.float64)
lib.c.smhip_fill(C.c_int(1), C.c_void_p(a.ptr), one.ctypes.data_as(C.c_void_p), C.c_size_t(2 * n))
lib.c.smhip_fill(C.c_int(1), C.c_void_p(b.ptr), one.ctypes.data_as(C.c_void_p), C.c_size_t(2 * n))
sp = lib.alloc(16)
t = timeit(lambda: lib.dot_c64_async(a.ptr, b.ptr, n, sp))
print("%-34s %10.1f %9.0f %6.1f%%" % ("complex128 dot n=2^26", t, 32.0 * n / t * 1e-3, 32.0 * n / t * 1e-3 / 80))
for _ in range(3): lib.dot_c64(a.ptr, b.ptr, n)
t0 = time.perf_counter()
for _ in range(20): r = lib.dot_c64(a.ptr, b.ptr, n)
t = (time.perf_counter() - t0) / 20 * 1e6
print("%-34s %10.1f %9.0f %6.1f%%   (value %s)" % ("  with the read-back (host clock)", t, 32.0 * n / t * 1e-3, 32.0 * n / t * 1e-3 / 80, r))
# the same at sizes the Infinity Cache holds (plain-load policy)
for lg in (20, 22):
    m = 1 << lg
    t = timeit(lambda: lib.dot_c64_async(a.ptr, b.ptr, m, sp), steps=200)
    print("%-34s %10.1f %9.0f %6.1f%%" % ("complex128 dot n=2^%d" % lg, t, 32.0 * m / t * 1e-3, 32.0 * m / t * 1e-3 / 80))
lib.free(sp)
