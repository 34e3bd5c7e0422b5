"""Is the 2R+1W stream sensitive to the relative placement of a, b, c?  (bimodal 6.27 / 6.54 TB/s across runs)"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
GiB = 1 << 30
def timeit(args, steps=40):
    fn = lib.c.smhip_contiguous
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps
slab = lib.alloc(3 * GiB + (256 << 20))
print("slab at", hex(slab), "slab %% 2MiB =", slab % (2 << 20))
lib.c.smhip_fill_uniform_f32.restype = C.c_int
lib.c.smhip_fill_uniform_f32(C.c_void_p(slab), C.c_size_t((3 * GiB + (256 << 20)) // 4), 1, 0, C.c_float(-1.0), C.c_float(1.0))
for db, dc in ((0, 0), (4096, 8192), (1 << 14, 1 << 15), (1 << 16, 1 << 17), (1 << 18, 1 << 19), (1 << 20, 1 << 21), (1 << 22, 1 << 23),
               (1 << 24, 1 << 25), (3 << 20, 7 << 20), (5 << 12, 11 << 12), (0, 0), (1 << 20, 1 << 21)):
    a, b, c = slab, slab + GiB + db, slab + 2 * GiB + dc
    ms = timeit((C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), C.c_size_t(n)))
    print("b +%-9d c +%-9d  %.4f ms  %.0f GB/s" % (db, dc, ms, 12.0 * n / ms * 1e-6))
# separate allocations, as bench.py makes them
x, y, z = lib.alloc(GiB), lib.alloc(GiB), lib.alloc(GiB)
print("separate allocs:", hex(x), hex(y), hex(z), "deltas", hex(y - x), hex(z - y))
ms = timeit((C.c_int(0), C.c_int(0), C.c_void_p(x), C.c_void_p(y), C.c_void_p(z), C.c_size_t(n)))
print("separate allocs  %.4f ms  %.0f GB/s" % (ms, 12.0 * n / ms * 1e-6))
