import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
for dt in (np.float32, np.float64):
    n = (1 << 30) // np.dtype(dt).itemsize
    a = lib.empty((n,), dt)
    v = np.array([1], dtype=dt)
    args = (C.c_int(sma.DTYPES[np.dtype(dt)]), C.c_void_p(a.ptr), v.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    for _ in range(5): lib.c.smhip_fill(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(50): lib.c.smhip_fill(*args)
    lib.record(e1); lib.synchronize()
    t = lib.elapsed_ms(e0, e1) / 50 * 1000
    print("fill %s 1 GiB: %.1f us  %.0f GB/s  %.1f%%" % (np.dtype(dt).name, t, (1 << 30) / t * 1e-3, (1 << 30) / t * 1e-3 / 80))
