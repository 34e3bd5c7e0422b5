"""Does pow(a, 2.5)'s time depend on the DATA?  (LDS table lookups: same index in every lane vs random indices)"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
n = 1 << 26
def t(a, y=2.5, steps=300):
    out = lib.empty((n,), np.float32)
    v = np.array([y], dtype=np.float32)
    args = (C.c_int(4), C.c_int(0), C.c_void_p(a.ptr), v.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_void_p(out.ptr))
    fn = lib.c.smhip_array_scalar
    for _ in range(2000): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
rnd = lib.uniform_f32(n, 5, 0.01, 100.0)
narrow = lib.uniform_f32(n, 5, 1.70, 1.71)
const = lib.empty((n,), np.float32)
lib.c.smhip_fill(C.c_int(0), C.c_void_p(const.ptr), np.array([1.7], dtype=np.float32).ctypes.data_as(C.c_void_p), C.c_size_t(n))
print("random (0.01, 100): %.1f us" % t(rnd))
print("narrow (1.70, 1.71): %.1f us" % t(narrow))
print("constant 1.7: %.1f us" % t(const))
wide = lib.uniform_f32(n, 9, 1e-30, 1e30)
print("uniform (1e-30, 1e30) [still mostly one binade]: %.1f us" % t(wide))
