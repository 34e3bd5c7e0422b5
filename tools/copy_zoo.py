"""Assignments dst[...] = src[...] through smhip_copy_strided (SMArray's `view = array`), f32, 128-256 MiB: % of 8 B per element moved.
    python tools/copy_zoo.py"""
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
def dense(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc); acc *= d
    return st[::-1]
def run(name, shape, ss, sd, nsrc, ndst, soff=0, doff=0):
    n = int(np.prod(shape))
    src = lib.uniform_f32(nsrc, 1, -1, 1); dst = lib.uniform_f32(ndst, 2, -1, 1)
    fn = lambda: lib.c.smhip_copy_strided(C.c_int(0), C.c_void_p(src.ptr + 4 * soff), i64(ss), C.c_void_p(dst.ptr + 4 * doff), i64(sd), i64(shape), C.c_int(len(shape)))
    t = timeit(fn)
    print("%-56s n %10d  %8.1f us  %5.1f %%" % (name, n, t, 8.0 * n / t * 1e-3 / 80), flush=True)
    del src, dst; lib.pool_trim()
N = 8192
run("dst = src (dense)", (N, N), (N, 1), (N, 1), N * N, N * N)
run("dst = src.T", (N, N), (1, N), (N, 1), N * N, N * N)
run("dst.T = src", (N, N), (N, 1), (1, N), N * N, N * N)
run("dst = src.T at 12288^2 (the wide patch)", (12288, 12288), (1, 12288), (12288, 1), 12288 * 12288, 12288 * 12288)
run("dst = src.T at 16384^2", (16384, 16384), (1, 16384), (16384, 1), 1 << 28, 1 << 28)
run("dst[:, :6144] = src", (N, 6144), (6144, 1), (N, 1), N * 6144, N * N)
run("dst[1:-1, 1:-1] = src", (N - 2, N - 2), (N - 2, 1), (N, 1), (N - 2) * (N - 2), N * N, doff=N + 1)
run("dst[::2, :] = src", (N // 2, N), (N, 1), (2 * N, 1), N * N // 2, N * N)
run("dst[:, ::2] = src", (N, N // 2), (N // 2, 1), (N, 2), N * N // 2, N * N)
run("dst[:, ::2] = src[:, ::2]", (N, N // 2), (N, 2), (N, 2), N * N, N * N)
run("dst = src[:, ::2]", (N, N // 2), (N, 2), (N // 2, 1), N * N, N * N // 2)
run("dst (n,3) = src (3,n).T", (1 << 24, 3), (1, 1 << 24), (3, 1), 3 << 24, 3 << 24)
run("dst (3,n) = src (n,3).T", (3, 1 << 24), (1, 3), (1 << 24, 1), 3 << 24, 3 << 24)
run("dst (n,3).T = src (3,n)  [dst view turned]", (3, 1 << 24), (1 << 24, 1), (1, 3), 3 << 24, 3 << 24)
run("dst[:, :, 0] = src  (one channel of (4096,4096,4))", (4096, 4096), (4096, 1), (16384, 4), 4096 * 4096, 4096 * 4096 * 4)
run("dst (B,C,HW) = src (B,HW,C) perm, C = 3", (256, 3, 50176), (150528, 1, 3), (150528, 50176, 1), 256 * 150528, 256 * 150528)
run("dst = src broadcast row (1, N) -> (N, N)", (N, N), (0, 1), (N, 1), N, N * N)
run("dst = src broadcast column (N, 1) -> (N, N)", (N, N), (1, 0), (N, 1), N, N * N)
run("dst (64,1024,1024) = src perm(0,2,1)", (64, 1024, 1024), (1 << 20, 1, 1024), dense((64, 1024, 1024)), 1 << 26, 1 << 26)
run("dst perm(0,2,1) = src (64,1024,1024)", (64, 1024, 1024), dense((64, 1024, 1024)), (1 << 20, 1, 1024), 1 << 26, 1 << 26)
