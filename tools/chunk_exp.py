"""Does cutting a 2^31-element add into 2^28-element launches recover the rate lost at very large sizes?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
n = 1 << 31
a = lib.uniform_f32(n, 1, -1.0, 1.0); b = lib.uniform_f32(n, 2, -1.0, 1.0); c = lib.empty((n,), np.float32)
fn = lib.c.smhip_contiguous
def run(chunk):
    for off in range(0, n, chunk):
        fn(C.c_int(0), C.c_int(0), C.c_void_p(a.ptr + 4 * off), C.c_void_p(b.ptr + 4 * off), C.c_void_p(c.ptr + 4 * off), C.c_size_t(chunk))
for lg in (31, 30, 29, 28, 26):
    chunk = 1 << lg
    for _ in range(3): run(chunk)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(10): run(chunk)
    lib.record(e1); lib.synchronize()
    ms = lib.elapsed_ms(e0, e1) / 10
    print("2^31 add in launches of 2^%d: %.3f ms  %.0f GB/s  %.1f%%" % (lg, ms, 12.0 * n / ms * 1e-6, 12.0 * n / ms * 1e-6 / 80), flush=True)
