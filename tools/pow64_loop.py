import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
n = 1 << 26
a = lib.empty((n,), np.float64); out = lib.empty((n,), np.float64)
h = np.random.default_rng(1).uniform(0.01, 100.0, 1 << 20)
for i in range(0, n, 1 << 20): lib.upload(a.ptr + i * 8, h)
v = np.array([2.5], dtype=np.float64)
args = (C.c_int(4), C.c_int(1), C.c_void_p(a.ptr), v.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_void_p(out.ptr))
for _ in range(300): lib.c.smhip_array_scalar(*args)
e0, e1 = lib.event(), lib.event()
lib.synchronize(); lib.record(e0)
for _ in range(100): lib.c.smhip_array_scalar(*args)
lib.record(e1); lib.synchronize()
t = lib.elapsed_ms(e0, e1) / 100 * 1000
print("f64 pow(a, 2.5) n=2^26: %.1f us  %.0f GB/s (%.1f%%)" % (t, 16.0 * n / t * 1e-3, 16.0 * n / t * 1e-3 / 80))
