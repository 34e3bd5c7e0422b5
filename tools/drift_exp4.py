"""Fresh process: three separate 1 GiB allocations (as bench.py makes them) vs one 4 GiB slab, first thing."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
GiB = 1 << 30
def rate(a, b, c, steps=60):
    fn = lib.c.smhip_contiguous
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), C.c_size_t(n))
    for _ in range(20): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return 12.0 * n / (lib.elapsed_ms(e0, e1) / steps) * 1e-6
def fill(p, nbytes, seed): lib.c.smhip_fill_uniform_f32(C.c_void_p(p), C.c_size_t(nbytes // 4), seed, 0, C.c_float(-1.0), C.c_float(1.0))
mode = sys.argv[1]
if mode == "sep":
    a = lib.alloc(GiB); fill(a, GiB, 1); b = lib.alloc(GiB); fill(b, GiB, 2); c = lib.alloc(GiB)
else:
    s = lib.alloc(4 * GiB); a, b, c = s, s + GiB, s + 2 * GiB; fill(a, GiB, 1); fill(b, GiB, 2)
print(mode, "%.0f" % rate(a, b, c))
