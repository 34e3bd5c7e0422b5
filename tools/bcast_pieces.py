"""Do the broadcast kernels gain from shorter launches too?  1 GiB-per-operand problems whole and cut along dim 0 (SMHIP_BCAST_PIECE_LOG2).
python tools/bcast_pieces.py   (run once per setting of the environment variable)"""
import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=12):
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    res = []
    for _ in range(3):
        lib.synchronize(); lib.record(e0)
        for _ in range(steps): fn()
        lib.record(e1); lib.synchronize()
        res.append(lib.elapsed_ms(e0, e1) / steps * 1000)
    return sorted(res)[1]
R = Cc = 16384
n = R * Cc
A = lib.uniform_f32(n, 1, -1, 1); B = lib.uniform_f32(n, 2, -1, 1); out = lib.empty((n,), np.float32); row = lib.uniform_f32(Cc, 3, -1, 1)
f32 = C.c_int(0)
def ew(op, a, sa, b, sb, shape): return lambda: lib.c.smhip_elementwise(C.c_int(op), f32, C.c_void_p(a.ptr), i64(sa), C.c_void_p(b.ptr), i64(sb), i64(shape), C.c_int(len(shape)), C.c_void_p(out.ptr))
tag = os.environ.get("SMHIP_BCAST_PIECE_LOG2", "whole")
for name, fn, byts in (("A.T + B        (tile kernel, 3 streams)", ew(0, A, [1, Cc], B, [Cc, 1], [Cc, R]), 12.0 * n),
                       ("A * row        (flat rows, 2 streams)", ew(2, A, [Cc, 1], row, [0, 1], [R, Cc]), 8.0 * n),
                       ("A[1:,1:]+B[1:,1:] (row kernel, 3 streams)", (lambda: lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(A.ptr + 4 * (Cc + 1)), i64([Cc, 1]), C.c_void_p(B.ptr + 4 * (Cc + 1)), i64([Cc, 1]), i64([R - 1, Cc - 1]), C.c_int(2), C.c_void_p(out.ptr))), 12.0 * (R - 1) * (Cc - 1))):
    t = timeit(fn)
    print("pieces=%-6s %-44s %9.1f us %5.1f %%" % (tag, name, t, byts / t * 1e-3 / 80), flush=True)
