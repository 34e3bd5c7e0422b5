"""A[:, ::2] = B[:, ::2] on (8192, 8192) f32 next to a dense copy of the same arrays, 40 launches each (for counter passes)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
M = 8192
n = M * M
a = lib.uniform_f32(n, 1, 0.5, 2.0); out = lib.uniform_f32(n, 2, 0.5, 2.0)
f32 = C.c_int(0)
def copy(ss, sd, shape): return lambda: lib.c.smhip_copy_strided(f32, C.c_void_p(a.ptr), i64(ss), C.c_void_p(out.ptr), i64(sd), i64(shape), C.c_int(len(shape)))
def timeit(fn, steps=40):
    for _ in range(5): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
print("dense copy (8192,8192)        %7.1f us" % timeit(copy((M, 1), (M, 1), (M, M))), flush=True)
print("A[:, ::2] = B[:, ::2]         %7.1f us" % timeit(copy((M, 2), (M, 2), (M, M // 2))), flush=True)
print("A[:, :4096] = B[:, :4096]     %7.1f us" % timeit(copy((M, 1), (M, 1), (M, M // 2))), flush=True)
