"""A.T + B through the LDS tile kernel at 4096^2 .. 16384^2 and at non-power-of-two pitches: rate per size (and the launches tools/pmc_tile_big.sh counts)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
print("library:", sys.argv[1] if len(sys.argv) > 1 else "built", flush=True)
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=10):
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    res = []
    for _ in range(3):
        lib.synchronize(); lib.record(e0)
        for _ in range(steps): fn()
        lib.record(e1); lib.synchronize()
        res.append(lib.elapsed_ms(e0, e1) / steps * 1000)
    return sorted(res)[1]
f32 = C.c_int(0)
for R, pitch in ((4096, 4096), (8192, 8192), (8192, 8256), (12288, 12288), (16384, 16384), (16384, 16448), (16000, 16000)):
    n = R * pitch
    A = lib.uniform_f32(n, 1, -1, 1); B = lib.uniform_f32(R * R, 2, -1, 1); out = lib.empty((R * R,), np.float32)
    # A is (R, R) inside rows of `pitch` elements; A.T has strides (1, pitch)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(A.ptr), i64([1, pitch]), C.c_void_p(B.ptr), i64([R, 1]), i64([R, R]), C.c_int(2), C.c_void_p(out.ptr))
    t = timeit(fn)
    print("A.T + B  %5d x %5d  pitch %5d  %9.1f us  %5.1f %%" % (R, R, pitch, t, 12.0 * R * R / t * 1e-3 / 80), flush=True)
    del A, B, out; lib.pool_trim()

# two turned operands, and cold operands (four operand sets in rotation: 3 GiB at 8192^2)
for R in (8192, 16384):
    A = lib.uniform_f32(R * R, 1, -1, 1); B = lib.uniform_f32(R * R, 2, -1, 1); out = lib.empty((R * R,), np.float32)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(A.ptr), i64([1, R]), C.c_void_p(B.ptr), i64([1, R]), i64([R, R]), C.c_int(2), C.c_void_p(out.ptr))
    t = timeit(fn)
    print("A.T + B.T %5d x %5d              %9.1f us  %5.1f %%" % (R, R, t, 12.0 * R * R / t * 1e-3 / 80), flush=True)
    del A, B, out; lib.pool_trim()
R = 8192
sets = [(lib.uniform_f32(R * R, 1 + k, -1, 1), lib.uniform_f32(R * R, 9 + k, -1, 1), lib.empty((R * R,), np.float32)) for k in range(4)]
state = [0]
def rot():
    A, B, out = sets[state[0] % 4]; state[0] += 1
    lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(A.ptr), i64([1, R]), C.c_void_p(B.ptr), i64([R, 1]), i64([R, R]), C.c_int(2), C.c_void_p(out.ptr))
t = timeit(rot, 12)
print("A.T + B   %5d x %5d  cold (4 sets) %9.1f us  %5.1f %%" % (R, R, t, 12.0 * R * R / t * 1e-3 / 80), flush=True)
