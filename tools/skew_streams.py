"""Does the relative placement of the three streams of out = a + b matter?  N = 2^28 f32 (1 GiB each) carved out of one
allocation with different byte skews between the streams (16-byte multiples)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
n = 1 << 28
GiB = 1 << 30
slack = 64 << 20
base = lib.alloc(3 * GiB + 3 * slack)
lib.c.smhip_fill_uniform_f32(C.c_void_p(base), C.c_size_t((3 * GiB + 3 * slack) // 4), C.c_uint64(1), C.c_uint64(0), C.c_float(-1.0), C.c_float(1.0))
def timeit(a, b, o, steps=30):
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(o), C.c_size_t(n))
    for _ in range(5): lib.c.smhip_contiguous(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): lib.c.smhip_contiguous(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize()
print("skew of b / of out (bytes)          us      % of 8 TB/s")
for sb, so in ((0, 0), (256, 512), (1024, 2048), (4096, 8192), (4096 + 256, 8192 + 512), (16384, 32768), (65536, 131072), (1 << 20, 2 << 20), ((1 << 20) + 4096, (2 << 20) + 8192),
               (16 << 20, 32 << 20), ((16 << 20) + 65536 + 4096 + 256, (32 << 20) + 131072 + 8192 + 512), (2 << 20, 4 << 20), (768, 1792), (12288, 28672)):
    t = timeit(base, base + GiB + slack + sb, base + 2 * (GiB + slack) + so)
    print("%10d / %-10d       %8.1f   %5.1f%%" % (sb, so, t, 12.0 * n / t * 1e-3 / 80), flush=True)
