"""f64 pow(a, s) at N = 2^26, random bases, for every s = -8 ... 8 in steps of one half: the double-double product chain
(sm_pow64.h: pow_halfint) against the general exp(s log a) form (SMHIP_POW_HALFINT_MAX=0 switches the chain off).
    python tools/pow64_halfint.py            one column; run it twice, with and without the switch"""
import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def timeit(fn, args, steps=30):
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
n = 1 << 26
rng = np.random.default_rng(5)
a = lib.to_device(rng.uniform(0.01, 100.0, n)); out = lib.empty((n,), np.float64)
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize()
print("SMHIP_POW_HALFINT_MAX =", os.environ.get("SMHIP_POW_HALFINT_MAX", "(unset)"))
for m2 in range(-16, 17):
    if m2 == 0: continue
    sy = C.c_double(m2 * 0.5)
    t = timeit(lib.c.smhip_array_scalar, (C.c_int(4), C.c_int(1), C.c_void_p(a.ptr), C.byref(sy), C.c_size_t(n), C.c_void_p(out.ptr)))
    print("f64 pow(a, %5.1f)  %6.1f us  %5.1f %%" % (m2 * 0.5, t, 16.0 * n / t * 1e-3 / 80), flush=True)
