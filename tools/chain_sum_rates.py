"""sum(pow(A - B, 2)) -- the squared error -- as ONE pass (smhip_chain_sum_async: 8 B/elem) against the chain and the sum of its result
(smhip_chain + smhip_sum_async: 12 + 4 B/elem) and against the three operator calls (sub, pow, sum: 12 + 8 + 4); f32, events."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
c = lib.c
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def events(fn, steps):
    for _ in range(10): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
sp = lib.alloc(16)
print("%-12s %10s %22s %26s %30s" % ("shape", "elements", "one pass (8 B/elem)", "chain + sum (16 B/elem)", "sub, pow, sum (24 B/elem)"))
for rows, cols in ((2048, 4096), (4096, 4096), (8192, 8192), (16384, 8192)):
    n = rows * cols
    A = lib.uniform_f32(n, 1, -1, 1); B = lib.uniform_f32(n, 2, -1, 1); t1 = lib.empty((n,), np.float32); t2 = lib.empty((n,), np.float32)
    ptrs = (C.c_void_p * 3)(A.ptr, B.ptr, None)
    strides = i64([1, 1, 0]); scal = np.array([0, 0, 2], dtype=np.float32)
    ops = (C.c_int * 2)(sma.OP_SUB, sma.OP_POW); swp = (C.c_int * 2)(0, 0); shape = i64([n])
    args = (C.c_int(0), C.c_int(3), ptrs, strides, scal.ctypes.data_as(C.c_void_p), ops, swp, shape, C.c_int(1))
    fused = lambda: c.smhip_chain_sum_async(*args, C.c_void_p(sp))
    def two():
        c.smhip_chain(*args, C.c_void_p(t1.ptr)); c.smhip_sum_async(C.c_int(0), C.c_void_p(t1.ptr), C.c_size_t(n), C.c_void_p(sp))
    twof = np.float32(2)
    def three():
        c.smhip_contiguous(C.c_int(sma.OP_SUB), C.c_int(0), C.c_void_p(A.ptr), C.c_void_p(B.ptr), C.c_void_p(t1.ptr), C.c_size_t(n))
        c.smhip_array_scalar(C.c_int(sma.OP_POW), C.c_int(0), C.c_void_p(t1.ptr), C.byref(C.c_float(2.0)), C.c_size_t(n), C.c_void_p(t2.ptr))
        c.smhip_sum_async(C.c_int(0), C.c_void_p(t2.ptr), C.c_size_t(n), C.c_void_p(sp))
    steps = 100 if n <= (1 << 26) else 40
    tf, t2_, t3 = events(fused, steps), events(two, steps), events(three, steps)
    print("%-12s %10d %12.1f us %5.1f%% %16.1f us %5.1f%% %20.1f us %5.1f%%" % ("%dx%d" % (rows, cols), n, tf, 8.0 * n / tf * 1e-3 / 80, t2_, 16.0 * n / t2_ * 1e-3 / 80,
                                                                                   t3, 24.0 * n / t3 * 1e-3 / 80), flush=True)
    del A, B, t1, t2; lib.pool_trim()
