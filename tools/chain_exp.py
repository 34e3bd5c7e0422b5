"""Chains of dependent operators on mid-size arrays: does keeping the streams cacheable (no `nt`) let the next
operator read its input from the 256 MiB Infinity Cache / L2?   usage: chain_exp.py [path/to/libsmhip.so]"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
fn = lib.c.smhip_contiguous
def chain(a, b, t1, t2, t3, n):
    fn(C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(t1), C.c_size_t(n))   # t1 = a + b
    fn(C.c_int(2), C.c_int(0), C.c_void_p(t1), C.c_void_p(a), C.c_void_p(t2), C.c_size_t(n))  # t2 = t1 * a
    fn(C.c_int(1), C.c_int(0), C.c_void_p(t2), C.c_void_p(b), C.c_void_p(t3), C.c_size_t(n))  # t3 = t2 - b
print("%-10s %10s %12s %10s" % ("n", "MiB/array", "us/chain", "GB/s"))
for lg in range(18, 29):
    n = 1 << lg
    a = lib.uniform_f32(n, 1, -1.0, 1.0); b = lib.uniform_f32(n, 2, -1.0, 1.0)
    t = [lib.empty((n,), np.float32) for _ in range(3)]
    reps = max(20, min(2000, (1 << 31) // n // 12))
    for _ in range(reps // 4 + 1): chain(a.ptr, b.ptr, t[0].ptr, t[1].ptr, t[2].ptr, n)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(reps): chain(a.ptr, b.ptr, t[0].ptr, t[1].ptr, t[2].ptr, n)
    lib.record(e1); lib.synchronize()
    us = lib.elapsed_ms(e0, e1) / reps * 1000
    print("2^%-8d %10.1f %12.2f %10.0f" % (lg, n * 4 / 2**20, us, 36.0 * n / us * 1e-3), flush=True)
    del a, b, t
