"""Slab (one 4 GiB allocation carved into a, b, c) vs three separate 1 GiB allocations, re-allocated 8 times."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
GiB = 1 << 30
def rate(a, b, c, steps=30):
    fn = lib.c.smhip_contiguous
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), C.c_size_t(n))
    for _ in range(3): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return 12.0 * n / (lib.elapsed_ms(e0, e1) / steps) * 1e-6
def fill(p, nbytes): lib.c.smhip_fill_uniform_f32(C.c_void_p(p), C.c_size_t(nbytes // 4), 1, 0, C.c_float(-1.0), C.c_float(1.0))
for trial in range(8):
    sep = [lib.alloc(GiB) for _ in range(3)]
    for p in sep: fill(p, GiB)
    r_sep = [rate(sep[0], sep[1], sep[2]), rate(sep[2], sep[1], sep[0]), rate(sep[1], sep[2], sep[0])]
    for p in sep: lib.free(p)
    lib.pool_trim()
    junk = [lib.alloc((37 + 61 * trial) << 20), lib.alloc((900 + 113 * trial) << 20)]   # perturb the driver's free lists
    slab = lib.alloc(4 * GiB); fill(slab, 3 * GiB)
    r_slab = [rate(slab, slab + GiB, slab + 2 * GiB), rate(slab + 2 * GiB, slab + GiB, slab), rate(slab + GiB, slab + 2 * GiB, slab)]
    lib.free(slab)
    for p in junk: lib.free(p)
    lib.pool_trim()
    print("trial %d  separate (abc, cba, bca): %s   slab: %s" % (trial, " ".join("%.0f" % r for r in r_sep), " ".join("%.0f" % r for r in r_slab)))
