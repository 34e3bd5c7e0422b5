"""Broadcast forms beyond tools/bcast_matrix.py (out = x op y with zero strides on either side, f32): looks for weak kernels.
    python tools/bcast_zoo.py"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
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
def bstrides(shape, full):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc); acc *= d
    st = st[::-1]
    return [0 if d == 1 and f != 1 else s for d, s, f in zip(shape, st, full)]
def run(name, xs, ys, op=2):
    full = [max(a, b) for a, b in zip(xs, ys)]
    n = int(np.prod(full)); nx = int(np.prod(xs)); ny = int(np.prod(ys))
    x = lib.uniform_f32(nx, 1, 0.5, 2); y = lib.uniform_f32(ny, 2, 0.5, 2); out = lib.empty((n,), np.float32)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(op), C.c_int(0), C.c_void_p(x.ptr), i64(bstrides(xs, full)), C.c_void_p(y.ptr), i64(bstrides(ys, full)), i64(full), C.c_int(len(full)), C.c_void_p(out.ptr))
    t = timeit(fn)
    alg = 4.0 * (n + nx + ny)
    print("%-52s n %10d  %8.1f us  %5.1f %% of algorithmic" % (name, n, t, alg / t * 1e-3 / 80), flush=True)
    del x, y, out; lib.pool_trim()
run("(64,1024,1)+(64,1,1024) batched outer", (64, 1024, 1), (64, 1, 1024))
run("(8192,8191)*(1,8191) odd row", (8192, 8191), (1, 8191))
run("(8192,8191)*(8192,1) odd column", (8192, 8191), (8192, 1))
run("(64,256,56,56)+(1,256,1,1) NCHW channel bias", (64, 256, 56, 56), (1, 256, 1, 1))
run("(64,256,56,56)*(64,256,1,1) NCHW sample-channel scale", (64, 256, 56, 56), (64, 256, 1, 1))
run("(256,3,224,224)-(1,3,1,1) NCHW mean", (256, 3, 224, 224), (1, 3, 1, 1), op=1)
run("(256,3,224,224)/(1,3,224,224) per-pixel", (256, 3, 224, 224), (1, 3, 224, 224), op=3)
run("(1,)+(2^26,) scalar array", (1,), (1 << 26,), op=0)
run("(8,2^23)+(1,2^23) few long rows", (8, 1 << 23), (1, 1 << 23), op=0)
run("(8,2^23)+(8,1) few long rows, column", (8, 1 << 23), (8, 1), op=0)
run("(2^23,8)+(1,8) tiny rows, row", (1 << 23, 8), (1, 8), op=0)
run("(2^23,8)*(2^23,1) tiny rows, column", (1 << 23, 8), (1 << 23, 1))
run("(4096,1,4096)+(1,4,1) middle axis", (4096, 1, 4096), (1, 4, 1), op=0)
run("(128,1,512,1)+(1,128,1,8) alternating", (128, 1, 512, 1), (1, 128, 1, 8), op=0)
run("(32,2048,1,64)+(1,1,16,64)", (32, 2048, 1, 64), (1, 1, 16, 64), op=0)
run("(16,1,1024,1024)+(16,4,1,1024)", (16, 1, 1024, 1024), (16, 4, 1, 1024), op=0)
