"""Reductions over very large f32 arrays (2^28 .. 2^31 elements): sum, dot, fused add+sum -- rates per size (round 3: launched in pieces)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
def timeit(fn, steps):
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
for lg in (28, 30, 31):
    n = 1 << lg
    a = lib.uniform_f32(n, 6, 0.0, 1.0); b = lib.uniform_f32(n, 7, 0.0, 1.0); c = lib.empty((n,), np.float32)
    sp = lib.alloc(8)
    steps = 20 if lg == 28 else 6
    t = timeit(lambda: lib.sum_async(a, sp), steps); print("f32 sum            n=2^%d %9.1f us %5.1f%%" % (lg, t, 4.0 * n / t * 1e-3 / 80), flush=True)
    t = timeit(lambda: lib.dot_async(a, b, sp), steps); print("f32 dot            n=2^%d %9.1f us %5.1f%%" % (lg, t, 8.0 * n / t * 1e-3 / 80), flush=True)
    t = timeit(lambda: lib.contiguous_sum_async(sma.OP_ADD, a, b, c, sp), steps); print("f32 fused add+sum  n=2^%d %9.1f us %5.1f%%   sum %.6f" % (lg, t, 12.0 * n / t * 1e-3 / 80, lib.read_f64(sp)), flush=True)
    lib.free(sp); del a, b, c; lib.pool_trim()
