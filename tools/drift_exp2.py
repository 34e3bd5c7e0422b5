"""Which property of a placement decides the add's rate?  12 fresh 1 GiB buffers; triples and self-combinations."""
import sys, ctypes as C, itertools
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
def rate(a, b, c, steps=30):
    fn = lib.c.smhip_contiguous
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), C.c_size_t(n))
    for _ in range(3): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return 12.0 * n / (lib.elapsed_ms(e0, e1) / steps) * 1e-6
def churn():
    # allocate and free a pile of odd-sized blocks through the driver, like an earlier workload would have
    ps = [lib.alloc(sz) for sz in (3 << 20, 700 << 20, 5 << 20, 1500 << 20, 64 << 20, 2300 << 20, 9 << 20, 11 << 20)]
    for p in ps[::2]: lib.free(p)
    lib.pool_trim()
    for p in ps[1::2]: lib.free(p)
    lib.pool_trim()
bufs = [lib.alloc(1 << 30) for _ in range(6)]
for p in bufs: lib.c.smhip_fill_uniform_f32(C.c_void_p(p), C.c_size_t(n), 1, 0, C.c_float(-1.0), C.c_float(1.0))
print("fresh: ", [hex(p) for p in bufs])
print("fresh triples:", " ".join("%d%d%d:%.0f" % (i, j, k, rate(bufs[i], bufs[j], bufs[k])) for i, j, k in ((0, 1, 2), (3, 4, 5), (0, 2, 4), (1, 3, 5), (5, 4, 3), (0, 1, 5))))
for p in bufs: lib.free(p)
lib.pool_trim()
churn()
bufs = [lib.alloc(1 << 30) for _ in range(6)]
for p in bufs: lib.c.smhip_fill_uniform_f32(C.c_void_p(p), C.c_size_t(n), 1, 0, C.c_float(-1.0), C.c_float(1.0))
print("after churn:", [hex(p) for p in bufs])
print("churned triples:", " ".join("%d%d%d:%.0f" % (i, j, k, rate(bufs[i], bufs[j], bufs[k])) for i, j, k in ((0, 1, 2), (3, 4, 5), (0, 2, 4), (1, 3, 5), (5, 4, 3), (0, 1, 5))))
# in-place forms isolate single buffers: c = a + a (1R1W per buffer pair), and a = a + a
print("single-buffer in-place a=a+a:", " ".join("%d:%.0f" % (i, rate(bufs[i], bufs[i], bufs[i]) * 2 / 3) for i in range(6)), "(GB/s at 8 B/elem)")
