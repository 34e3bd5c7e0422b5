"""Does the pow / add kernel time depend on the relative placement of its input and output buffers?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def timeit(fn, args, steps=100):
    for _ in range(10): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
n = 1 << 26
a = lib.uniform_f32(n, 5, 0.01, 100.0)
big = lib.alloc(n * 4 + (64 << 20))
e = C.c_float(2.5)
print("a at", hex(a.ptr), "out base", hex(big), "delta base", hex(big - a.ptr))
for delta in (0, 256, 1024, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, (1 << 24) + (1 << 12), 1 << 25):
    t = timeit(lib.c.smhip_array_scalar, (C.c_int(4), C.c_int(0), C.c_void_p(a.ptr), C.byref(e), C.c_size_t(n), C.c_void_p(big + delta)))
    print("pow  out = base + %-10d %.1f us" % (delta, t))
# same question for the 2R+1W add at 2^26 (smaller, so placement effects are visible) and 2^28
b = lib.uniform_f32(n, 6, -1.0, 1.0)
for delta in (0, 4096, 65536, 1 << 20, 1 << 22, 1 << 24):
    t = timeit(lib.c.smhip_contiguous, (C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(big + delta), C.c_size_t(n)))
    print("add26 out = base + %-10d %.1f us  %.0f GB/s" % (delta, t, 12.0 * n / t * 1e-3))
