"""Separate 1 GiB allocations vs one slab carved into a, b, c -- alternating rounds in one process."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
GiB = 1 << 30
def timeit(a, b, c, steps=40):
    fn = lib.c.smhip_contiguous
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), C.c_size_t(n))
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    ms = lib.elapsed_ms(e0, e1) / steps
    return 12.0 * n / ms * 1e-6
def fill(p, nbytes):
    lib.c.smhip_fill_uniform_f32(C.c_void_p(p), C.c_size_t(nbytes // 4), 1, 0, C.c_float(-1.0), C.c_float(1.0))
sep = [lib.alloc(GiB) for _ in range(3)]
for p in sep: fill(p, GiB)
slab = lib.alloc(3 * GiB); fill(slab, 3 * GiB)
sep2 = [lib.alloc(GiB) for _ in range(3)]
for p in sep2: fill(p, GiB)
slab4 = lib.alloc(4 * GiB); fill(slab4, 4 * GiB)
print("sep ", [hex(p) for p in sep]); print("slab", hex(slab)); print("sep2", [hex(p) for p in sep2]); print("slab4", hex(slab4))
for r in range(4):
    print("round %d: separate %.0f | slab %.0f | separate#2 %.0f | slab4 %.0f | mixed(a,b sep; c slab) %.0f | slab reversed (c,b,a) %.0f" % (
        r, timeit(*sep), timeit(slab, slab + GiB, slab + 2 * GiB), timeit(*sep2), timeit(slab4, slab4 + GiB, slab4 + 2 * GiB),
        timeit(sep[0], sep[1], slab), timeit(slab + 2 * GiB, slab + GiB, slab)))
