import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def timeit(fns, steps=60):
    k = len(fns)
    for i in range(10): fns[i % k]()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for i in range(steps): fns[i % k]()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
one = np.array([1.5], dtype=np.float32); s = C.c_float(1.0000001)
for mib in (16, 32, 64, 128):
    n = mib << 18
    a = lib.empty((n,), np.float32); c = lib.empty((n,), np.float32)
    fill = lambda: lib.c.smhip_fill(C.c_int(0), C.c_void_p(a.ptr), one.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    mul = lambda: lib.c.smhip_array_scalar(C.c_int(2), C.c_int(0), C.c_void_p(a.ptr), C.byref(s), C.c_size_t(n), C.c_void_p(c.ptr))
    tf = timeit([fill]); tm = timeit([mul]); tb = timeit([fill, mul], steps=120) * 2
    print("%4d MiB: fill %6.1f us (%5.1f%%)  a*s alone %6.1f us   fill then a*s %6.1f us (sum of the two alone: %6.1f)" % (mib, tf, 4.0 * n / tf * 1e-3 / 80, tm, tb, tf + tm), flush=True)
