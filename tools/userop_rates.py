"""Run-time-compiled flat kernels against their built-in counterparts at N = 2^28 f32 (1 GiB per array) and 2^24."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
def timeit(fn, steps=30):
    for _ in range(5): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
op = lib.register_op("a + b")
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize(); del x
for lg in (28, 24):
    n = 1 << lg
    a = lib.uniform_f32(n, 1, 0.5, 2.0); b = lib.uniform_f32(n, 2, 0.5, 2.0); c = lib.uniform_f32(n, 3, 0.5, 2.0); d = lib.uniform_f32(n, 4, 0.5, 2.0); out = lib.empty((n,), np.float32)
    for name, fn, byts in (("built-in a + b", lambda: lib.contiguous(sma.OP_ADD, a, b, out), 12.0 * n), ("user Op a + b (hipRTC)", lambda: lib.contiguous(op, a, b, out), 12.0 * n),
                           ("built-in a * 2.5", lambda: lib.array_scalar(sma.OP_MUL, a, 2.5, out), 8.0 * n), ("user Op a + 2.5 (hipRTC)", lambda: lib.array_scalar(op, a, 2.5, out), 8.0 * n),
                           ("expr a0 + a1", lambda: lib.fused_expr("a0 + a1", a, b, out=out), 12.0 * n), ("expr (a0 + a1) * a2", lambda: lib.fused_expr("(a0 + a1) * a2", a, b, c, out=out), 16.0 * n),
                           ("built-in fused (a + b) * c", lambda: lib.fused(sma.OP_ADD, sma.OP_MUL, a, b, c, out), 16.0 * n), ("expr (a0 + a1) * a2 - a3", lambda: lib.fused_expr("(a0 + a1) * a2 - a3", a, b, c, d, out=out), 20.0 * n)):
        t = timeit(fn)
        print("2^%d  %-32s %8.1f us  %5.1f%%" % (lg, name, t, byts / t * 1e-3 / 80), flush=True)
    del a, b, c, d, out; lib.pool_trim()
