"""Batches of small transposed planes: out (B, n, n) = x(B, n, n) with its last two axes exchanged + dense y, f32.   python tools/small_planes.py"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
F64 = len(sys.argv) > 1 and sys.argv[1] == "f64"   # doubles: the arrays are raw memory of twice the size, the values do not matter
DTC, ESZ = (1, 8) if F64 else (0, 4)
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
for n, m in ((4, 4), (8, 8), (12, 12), (16, 16), (24, 24), (32, 32), (48, 48), (64, 64), (96, 96), (100, 100), (128, 128), (16, 64), (64, 16), (8, 128), (128, 8), (3, 224)):
    B = (1 << (25 if F64 else 26)) // (n * m)
    N = B * n * m
    x = lib.uniform_f32(N * ESZ // 4, 1, 1, 2); y = lib.uniform_f32(N * ESZ // 4, 2, 1, 2); out = lib.empty((N * ESZ // 4,), np.float32)
    # out (B, m, n): x is (B, n, m) read transposed: strides (n*m, 1, m)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(DTC), C.c_void_p(x.ptr), i64([n * m, 1, m]), C.c_void_p(y.ptr), i64([n * m, n, 1]), i64([B, m, n]), C.c_int(3), C.c_void_p(out.ptr))
    t = timeit(fn)
    print("B %8d planes of (%3d, %3d) -> (%3d, %3d)   %8.1f us  %5.1f %%" % (B, n, m, m, n, t, 3.0 * ESZ * N / t * 1e-3 / 80), flush=True)
    del x, y, out; lib.pool_trim()
