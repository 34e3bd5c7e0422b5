"""A zoo of view patterns beyond tools/bcast_matrix.py: out = view(x) + dense y, f32 unless noted, 128-512 MiB per array.  Looks for weak kernels.
    python tools/shape_zoo.py"""
import os, sys, ctypes as C, itertools
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
def strides_of(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc); acc *= d
    return st[::-1]
def run(name, base_shape, view_shape, view_strides, offset=0, dt=0, esz=4, streams=3):
    n = int(np.prod(view_shape)); nb = int(np.prod(base_shape))
    x = lib.uniform_f32(nb * esz // 4, 1, -1, 1); y = lib.uniform_f32(n * esz // 4, 2, -1, 1); out = lib.empty((n * esz // 4,), np.float32)
    ys = strides_of(view_shape)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(dt), C.c_void_p(x.ptr + offset * esz), i64(view_strides), C.c_void_p(y.ptr), i64(ys), i64(view_shape), C.c_int(len(view_shape)), C.c_void_p(out.ptr))
    t = timeit(fn)
    print("%-58s n %10d  %8.1f us  %5.1f %%" % (name, n, t, streams * esz * n / t * 1e-3 / 80), flush=True)
    del x, y, out; lib.pool_trim()
# 3-D permutations of (256, 512, 512) and a long-inner / short-inner variant
for shape in ((256, 512, 512), (64, 64, 16384), (16384, 64, 64), (1024, 1024, 64), (64, 1024, 1024)):
    st = strides_of(shape)
    for perm in itertools.permutations(range(3)):
        if perm == (0, 1, 2): continue
        run("%s perm%s" % (shape, perm), shape, [shape[p] for p in perm], [st[p] for p in perm])
# attention-style 4-D: (B, T, H, D) -> (B, H, T, D) and -> (B, H, D, T)
for shape in ((8, 2048, 32, 128), (8, 2048, 32, 64)):
    st = strides_of(shape)
    for perm in ((0, 2, 1, 3), (0, 2, 3, 1), (1, 0, 2, 3), (2, 0, 1, 3)):
        run("%s perm%s" % (shape, perm), shape, [shape[p] for p in perm], [st[p] for p in perm])
# slices
run("A[:, :6144] of (8192, 8192)", (8192, 8192), (8192, 6144), (8192, 1))
run("A[:, 1:6145] of (8192, 8192)", (8192, 8192), (8192, 6144), (8192, 1), offset=1)
run("A[1:8191, :] of (8192, 8192)", (8192, 8192), (8190, 8192), (8192, 1), offset=8192)
run("A[::3, :] of (12288, 8192)", (12288, 8192), (4096, 8192), (3 * 8192, 1))
run("A[:, ::3] of (4096, 12288)", (4096, 12288), (4096, 4096), (12288, 3))
run("A[:, ::4] of (4096, 16384)", (4096, 16384), (4096, 4096), (16384, 4))
run("A[:, :, 0] of (4096, 4096, 4)", (4096, 4096, 4), (4096, 4096), (16384, 4))
run("f64 A[:, ::2] of (4096, 8192)", (4096, 8192 * 2), (4096, 4096), (8192, 2), dt=1, esz=8)
run("1-D a[::2]", (2 ** 27,), (2 ** 26,), (2,))
run("1-D a[5:5+2^26]", (2 ** 26 + 64,), (2 ** 26,), (1,), offset=5)
