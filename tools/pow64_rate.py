"""f64 pow over 2^26 elements with random bases (table lookups scatter over the LDS copies): scalar and array exponents."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
def timeit(fn, args, steps=40):
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
n = 1 << 26
rng = np.random.default_rng(5)
a = lib.to_device(rng.uniform(0.01, 100.0, n)); b = lib.to_device(rng.uniform(0.5, 3.0, n)); out = lib.empty((n,), np.float64)
ones = lib.full((n,), 1.5, np.float64)
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize()
s = C.c_double(2.5)
for name, src in (("random bases", a), ("all bases 1.5", ones)):
    t = timeit(lib.c.smhip_array_scalar, (C.c_int(4), C.c_int(1), C.c_void_p(src.ptr), C.byref(s), C.c_size_t(n), C.c_void_p(out.ptr)))
    print("f64 pow(a, 2.5)  %-14s 2^26: %6.1f us  %5.1f%% of 8 TB/s" % (name, t, 16.0 * n / t * 1e-3 / 80), flush=True)
    t = timeit(lib.c.smhip_contiguous, (C.c_int(4), C.c_int(1), C.c_void_p(src.ptr), C.c_void_p(b.ptr), C.c_void_p(out.ptr), C.c_size_t(n)))
    print("f64 pow(a, b)    %-14s 2^26: %6.1f us  %5.1f%%" % (name, t, 24.0 * n / t * 1e-3 / 80), flush=True)
# other scalar exponents: multiples of one half up to 8 take the double-double product chain, everything else the general form
for y in (2.7, 1.5, 3.0, -2.5, 7.5, -8.0, 8.5, 0.3333333333333333, -13.37, 37.75, -1000.5, 5000.0):  # SMHIP_POW_SCALAR_LEVEL=0: the general form for all
    sy = C.c_double(y)
    t = timeit(lib.c.smhip_array_scalar, (C.c_int(4), C.c_int(1), C.c_void_p(a.ptr), C.byref(sy), C.c_size_t(n), C.c_void_p(out.ptr)))
    print("f64 pow(a, %-6.4g) random bases  2^26: %6.1f us  %5.1f%%" % (y, t, 16.0 * n / t * 1e-3 / 80), flush=True)
t = timeit(lib.c.smhip_array_scalar, (C.c_int(2), C.c_int(1), C.c_void_p(a.ptr), C.byref(s), C.c_size_t(n), C.c_void_p(out.ptr)))
print("f64 a * 2.5                      2^26: %6.1f us  %5.1f%%" % (t, 16.0 * n / t * 1e-3 / 80))
