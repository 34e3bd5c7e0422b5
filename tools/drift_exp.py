"""Does the f32 add's rate drift inside one process (clock / power states) or only between processes?"""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
def mk():
    a = lib.uniform_f32(n, 1, -1.0, 1.0); b = lib.uniform_f32(n, 2, -1.0, 1.0); c = lib.empty((n,), np.float32)
    return a, b, c
def rate(a, b, c, steps=60):
    fn = lib.c.smhip_contiguous
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(c.ptr), C.c_size_t(n))
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return 12.0 * n / (lib.elapsed_ms(e0, e1) / steps) * 1e-6
t0 = time.time()
a, b, c = mk()
print("same buffers over time:", " ".join("%.0f" % rate(a, b, c) for _ in range(40)), "(%.1f s)" % (time.time() - t0))
for r in range(6):
    del a, b, c
    lib.pool_trim()          # really hipFree
    a, b, c = mk()
    print("re-allocated #%d @%x:" % (r, a.ptr), " ".join("%.0f" % rate(a, b, c) for _ in range(5)))
time.sleep(3.0)
print("after 3 s idle:", " ".join("%.0f" % rate(a, b, c) for _ in range(8)))
