"""A transposed operand against a broadcast one: out (P, Q) = A(Q, P).T op {row (1, Q), column (P, 1), one value}, f32.   python tools/turned_bcast.py"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=10):
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    res = []
    for _ in range(3):
        lib.synchronize(); lib.record(e0)
        for _ in range(steps): fn()
        lib.record(e1); lib.synchronize()
        res.append(lib.elapsed_ms(e0, e1) / steps * 1000)
    return sorted(res)[1]
for P, Q in ((12288, 12288), (16384, 16384), (8192, 32768)):
    n = P * Q
    A = lib.uniform_f32(n, 1, 0.5, 2); y = lib.uniform_f32(max(P, Q), 2, 0.5, 2); out = lib.empty((n,), np.float32)
    for name, ys in (("row (1,Q)", (0, 1)), ("column (P,1)", (1, 0)), ("one value", (0, 0))):
        for swapped in (False, True):
            a_args = (C.c_void_p(A.ptr), i64([1, P])); b_args = (C.c_void_p(y.ptr), i64(ys))
            first, second = (b_args, a_args) if swapped else (a_args, b_args)
            fn = lambda: lib.c.smhip_elementwise(C.c_int(2), C.c_int(0), first[0], first[1], second[0], second[1], i64([P, Q]), C.c_int(2), C.c_void_p(out.ptr))
            t = timeit(fn)
            print("out %5d x %5d = %s  %-14s %8.1f us  %5.1f %% of 8 B/elem" % (P, Q, "y * A.T" if swapped else "A.T * y", name, t, 8.0 * n / t * 1e-3 / 80), flush=True)
    del A, y, out; lib.pool_trim()
