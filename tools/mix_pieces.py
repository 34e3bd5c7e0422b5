"""One stream mix at one size, for tools/mix_pieces.sh: python tools/mix_pieces.py <scalar|sum|dot|add> <log2n>  (SMHIP_PIECE_LOG2VEC from the environment)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
kind, lg = sys.argv[1], int(sys.argv[2])
n = 1 << lg
a = lib.uniform_f32(n, 1, 0.5, 2.0)
b = lib.uniform_f32(n, 2, 0.5, 2.0) if kind in ("dot", "add") else None
o = lib.empty((n,), np.float32) if kind in ("scalar", "add") else None
sp = lib.alloc(8)
one = C.c_float(1.0000001)
fn = {"scalar": lambda: lib.c.smhip_array_scalar(C.c_int(2), C.c_int(0), C.c_void_p(a.ptr), C.byref(one), C.c_size_t(n), C.c_void_p(o.ptr)),
      "sum": lambda: lib.sum_async(a, sp), "dot": lambda: lib.dot_async(a, b, sp),
      "add": lambda: lib.c.smhip_contiguous(C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(o.ptr), C.c_size_t(n))}[kind]
byts = {"scalar": 8.0, "sum": 4.0, "dot": 8.0, "add": 12.0}[kind] * n
for _ in range(5): fn()
res = []
e0, e1 = lib.event(), lib.event()
steps = 12 if lg >= 30 else 24
for _ in range(3):
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    res.append(lib.elapsed_ms(e0, e1) / steps * 1000)
t = sorted(res)[1]
print("%-6s N=2^%d  %9.1f us  %5.1f %%" % (kind, lg, t, byts / t * 1e-3 / 80))
