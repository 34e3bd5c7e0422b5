"""(N,224,224,3) + (1,224,1,3) and (N,224,224,3) + (3,) for a range of N: the LDS kernel (SMHIP_PERIODIC_MIN_MIB=100000)
against the periodic route (SMHIP_PERIODIC_MIN_MIB=0) -- where does writing the period out start to pay?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=60):
    for _ in range(8): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize(); del x
for N in (16, 32, 64, 96, 128, 192, 256):
    shape = (N, 224, 224, 3); n = int(np.prod(shape))
    a = lib.uniform_f32(n, 1, 0.5, 2.0); y = lib.uniform_f32(672, 2, 0.5, 2.0); out = lib.empty((n,), np.float32)
    for name, sy in (("(1,224,1,3)", [0, 3, 0, 1]), ("(3,)", [0, 0, 0, 1])):
        fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), i64([224*224*3, 224*3, 3, 1]), C.c_void_p(y.ptr), i64(sy), i64(shape), C.c_int(4), C.c_void_p(out.ptr))
        t = timeit(fn)
        print("N=%3d (%5.1f MiB out) + %-12s %7.1f us  %5.1f%%" % (N, 4.0 * n / 2**20, name, t, 8.0 * n / t * 1e-3 / 80), flush=True)
    del a, y, out; lib.pool_trim()
