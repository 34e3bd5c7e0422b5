"""Column-slice views A[:, :w] of a (8192, 8192) f32 array (row pitch 32 KiB, a power of two) against dense arrays of the
same shape: does the row kernel camp on a few channels when only part of every pitched row is touched?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=60):
    for _ in range(8): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
M = 8192
A = lib.uniform_f32(M * M, 1, 0.5, 2.0); B = lib.uniform_f32(M * M, 2, 0.5, 2.0); out = lib.empty((M * M,), np.float32)
f32 = C.c_int(0)
print("%-46s %10s %8s" % ("case", "us", "% peak"))
for w in (256, 1024, 2048, 4096, 8192):
    for pitch_a, pitch_o, name in ((M, w, "A[:, :%d] + B[:, :%d] -> dense" % (w, w)), (w, w, "dense (8192,%d) + dense" % w), (M + 64, w, "the same views at pitch 8256")):
        if pitch_a == M + 64 and w == M: continue
        fn = lambda: lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(A.ptr), i64([pitch_a, 1]), C.c_void_p(B.ptr), i64([pitch_a, 1]), i64([M if pitch_a != M + 64 else M - 64, w]), C.c_int(2), C.c_void_p(out.ptr))
        rows = M if pitch_a != M + 64 else M - 64
        t = timeit(fn)
        print("%-46s %10.1f %7.1f%%" % (name, t, 12.0 * rows * w / t * 1e-3 / 80), flush=True)
