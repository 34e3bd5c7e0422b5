"""User-defined (hipRTC) Ops against the built-in kernels on the same shapes."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, args, steps=50):
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
user = lib.register_op("a * b")
user2 = lib.register_op("(a + b) * 2")
M = 8192
cases = [
    ("contiguous 2^26", (M * M,), (1,), (1,), 12),
    ("(8192,8192) * (1,8192) row", (M, M), (M, 1), (0, 1), 8),
    ("(8192,8192) * (8192,1) column", (M, M), (M, 1), (1, 0), 8),
    ("A.T * B (8192,8192)", (M, M), (1, M), (M, 1), 12),
    ("(256,224,224,3) * (1,224,1,3)", (256, 224, 224, 3), (224 * 224 * 3, 224 * 3, 3, 1), (0, 3, 0, 1), 8),
]
print("%-36s %10s %10s %10s" % ("shape", "builtin us", "user us", "ratio"))
for name, shape, sa, sb, bpe in cases:
    n = int(np.prod(shape))
    a = lib.uniform_f32(n, 1, 0.5, 2.0); b = lib.uniform_f32(n, 2, 0.5, 2.0); out = lib.empty((n,), np.float32)
    def args(op): return (C.c_int(op), C.c_int(0), C.c_void_p(a.ptr), i64(sa), C.c_void_p(b.ptr), i64(sb), i64(shape), C.c_int(len(shape)), C.c_void_p(out.ptr))
    tb = timeit(lib.c.smhip_elementwise, args(2)); tu = timeit(lib.c.smhip_elementwise, args(user))
    print("%-36s %10.1f %10.1f %10.2f   (user: %.0f GB/s)" % (name, tb, tu, tu / tb, bpe * n / tu * 1e-3), flush=True)
    del a, b, out
