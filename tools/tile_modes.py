"""The LDS tile kernel on (8192, 8192) f32 with one and with two turned operands: A.T + B and A.T + B.T, 60 launches each
(for rocprofv3 passes; prints HIP-event rates when run alone)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
M = 8192
n = M * M
a = lib.uniform_f32(n, 1, -1.0, 1.0); b = lib.uniform_f32(n, 2, -1.0, 1.0); out = lib.empty((n,), np.float32)
def run(sa, sb, steps=60):
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), i64(sa), C.c_void_p(b.ptr), i64(sb), i64((M, M)), C.c_int(2), C.c_void_p(out.ptr))
    for _ in range(5): lib.c.smhip_elementwise(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): lib.c.smhip_elementwise(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
for name, sa, sb in (("A.T + B  ", (1, M), (M, 1)), ("A.T + B.T", (1, M), (1, M)), ("A + B    ", (M, 1), (M, 1))):
    t = run(sa, sb)
    print("%s (8192,8192): %6.1f us  %5.1f%% of 8 TB/s" % (name, t, 12.0 * n / t * 1e-3 / 80), flush=True)
# the same two turned operands out of arrays whose row pitch is not a power of two (8192 + 64 elements)
P = M + 64
a = lib.uniform_f32(M * P, 1, -1.0, 1.0); b = lib.uniform_f32(M * P, 2, -1.0, 1.0)
t = run((1, P), (1, P))
print("A.T + B.T, operand pitch 8256: %6.1f us  %5.1f%%" % (t, 12.0 * n / t * 1e-3 / 80), flush=True)
t = run((1, P), (M, 1))
print("A.T + B,   operand pitch 8256: %6.1f us  %5.1f%%" % (t, 12.0 * n / t * 1e-3 / 80), flush=True)
