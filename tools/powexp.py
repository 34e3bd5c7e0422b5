import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
n = 1 << 26
def run(name, a, steps=200):
    out = lib.empty((n,), np.float32)
    e = C.c_float(2.5)
    fn = lib.c.smhip_array_scalar
    args = (C.c_int(4), C.c_int(0), C.c_void_p(a.ptr), C.byref(e), C.c_size_t(n), C.c_void_p(out.ptr))
    for _ in range(20): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize()
    lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    print("%-40s %.1f us" % (name, lib.elapsed_ms(e0, e1) / steps * 1000))
run("uniform(0.01,100) seed 5", lib.uniform_f32(n, 5, 0.01, 100.0))
run("constant 1.7", lib.full((n,), 1.7, np.float32))
run("uniform(1.0,2.0)", lib.uniform_f32(n, 5, 1.0, 2.0))
run("uniform(0.01,100) seed 5 again", lib.uniform_f32(n, 5, 0.01, 100.0))
run("uniform(0.01,100) 50 steps", lib.uniform_f32(n, 5, 0.01, 100.0), 50)
# sweep-tool-like data: 0.01 + k * 99.99/2^24 with k = (i*2654435761) & 0xffffff
i = np.arange(n, dtype=np.uint64)
k = (i * np.uint64(2654435761)) & np.uint64(0xffffff)
x = (np.float32(0.01) + k.astype(np.float32) * np.float32(99.99 / 16777216.0)).astype(np.float32)
run("sweep-tool pattern", lib.to_device(x))
