"""How fast are the broadcast shapes that fall to the gather kernel?  (channels-last bias: inner extent 3)"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*seq)
def timeit(fn, args, steps=200):
    for _ in range(20): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
cases = [
    ("(32,224,224,3)+(1,224,1,3)", (32, 224, 224, 3), (224 * 224 * 3, 224 * 3, 3, 1), (0, 3, 0, 1), 224 * 3),
    ("(32,224,224,3)+(1,1,1,3)", (32, 224, 224, 3), (224 * 224 * 3, 224 * 3, 3, 1), (0, 0, 0, 1), 3),
    ("(64,56,56,256)+(1,1,1,256)", (64, 56, 56, 256), (56 * 56 * 256, 56 * 256, 256, 1), (0, 0, 0, 1), 256),
    ("(4096,4096)+(4096,1)", (4096, 4096), (4096, 1), (1, 0), 4096),
    ("(8192,8192)+(1,8192)", (8192, 8192), (8192, 1), (0, 1), 8192),
]
for name, shape, sa, sb, nb in cases:
    n = int(np.prod(shape))
    a = lib.uniform_f32(n, 1, -1.0, 1.0); b = lib.uniform_f32(nb, 2, -1.0, 1.0); out = lib.empty((n,), np.float32)
    t = timeit(lib.c.smhip_elementwise, (C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), i64(sa), C.c_void_p(b.ptr), i64(sb), i64(shape), C.c_int(len(shape)), C.c_void_p(out.ptr)))
    print("%-32s n=%-10d %8.1f us  %7.0f GB/s (8 B/elem)" % (name, n, t, 8.0 * n / t * 1e-3))
