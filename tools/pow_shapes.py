"""f32 pow over 2^24 elements through the three entry points it can take (scalar exponent, array exponent, a row of
exponents broadcast over the rows), plus the same shapes with multiply: where does the row kernel's pow lose its time?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, args, steps=200):
    for _ in range(20): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
R = Cc = 4096
n = R * Cc
a = lib.uniform_f32(n, 1, 0.01, 100.0); b = lib.uniform_f32(n, 2, 0.5, 3.0); row = lib.uniform_f32(Cc, 3, 0.5, 3.0); out = lib.empty((n,), np.float32)
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize()
for opn, op in (("mul", 2), ("pow", 4)):
    s = C.c_float(2.5)
    t = timeit(lib.c.smhip_array_scalar, (C.c_int(op), C.c_int(0), C.c_void_p(a.ptr), C.byref(s), C.c_size_t(n), C.c_void_p(out.ptr)))
    print("%s  a op scalar            2^24: %6.1f us  %5.1f%% of 8 TB/s" % (opn, t, 8.0 * n / t * 1e-3 / 80))
    t = timeit(lib.c.smhip_contiguous, (C.c_int(op), C.c_int(0), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(out.ptr), C.c_size_t(n)))
    print("%s  a op b (contiguous)    2^24: %6.1f us  %5.1f%%" % (opn, t, 12.0 * n / t * 1e-3 / 80))
    t = timeit(lib.c.smhip_elementwise, (C.c_int(op), C.c_int(0), C.c_void_p(a.ptr), i64([Cc, 1]), C.c_void_p(row.ptr), i64([0, 1]), i64([R, Cc]), C.c_int(2), C.c_void_p(out.ptr)))
    print("%s  (4096,4096) op (1,4096)     : %6.1f us  %5.1f%%" % (opn, t, (8.0 * n + 4 * Cc) / t * 1e-3 / 80))
    t = timeit(lib.c.smhip_elementwise, (C.c_int(op), C.c_int(0), C.c_void_p(a.ptr), i64([Cc, 1]), C.c_void_p(row.ptr), i64([1, 0]), i64([R, Cc]), C.c_int(2), C.c_void_p(out.ptr)))
    print("%s  (4096,4096) op (4096,1)     : %6.1f us  %5.1f%%" % (opn, t, (8.0 * n + 4 * R) / t * 1e-3 / 80))
