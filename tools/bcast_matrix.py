"""Rates of smhip_elementwise over the broadcast / view shapes element_wise_op meets (f32 unless noted).

Algorithmic bytes = distinct operand elements read + output written (a broadcast operand counts once).
usage: python tools/bcast_matrix.py [> profiles/rNN_bcast_matrix.txt]
"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, args, steps):
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
def dense(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc); acc *= d
    return tuple(reversed(st))
def span(shape, st): return 1 + sum((d - 1) * s for d, s in zip(shape, st))
def distinct(shape, st):  # elements actually touched (stride-0 dims collapse)
    k = 1
    for d, s in zip(shape, st):
        if s != 0: k *= d
    return k

cases = []
def case(name, shape, sa, sb, off_a=0, off_b=0, op=0, dt=np.float32): cases.append((name, shape, sa, sb, off_a, off_b, op, dt))
M = 8192
case("dense (8192,8192) + same              [contig]", (M, M), dense((M, M)), dense((M, M)))
case("C3 (4096,4096) * (1,4096)              [flat]", (4096, 4096), (4096, 1), (0, 1), op=2)
case("(8192,8192) * (1,8192)                 [flat]", (M, M), (M, 1), (0, 1), op=2)
case("(8192,8192) * (8192,1) column          [flat]", (M, M), (M, 1), (1, 0), op=2)
case("(8192,1) * (1,8192) outer product      [row]", (M, M), (1, 0), (0, 1), op=2)
case("(64,56,56,256) + (1,1,1,256) bias      [flat]", (64, 56, 56, 256), dense((64, 56, 56, 256)), (0, 0, 0, 1))
case("(256,224,224,3) + (1,224,1,3) ref test [flat]", (256, 224, 224, 3), dense((256, 224, 224, 3)), (0, 3, 0, 1))
case("(128,224,224,3) + (1,224,1,3) ref test [lds]", (128, 224, 224, 3), dense((128, 224, 224, 3)), (0, 3, 0, 1))
case("(256,224,224,3) + (1,1,1,3) RGB bias   [flat]", (256, 224, 224, 3), dense((256, 224, 224, 3)), (0, 0, 0, 1))
case("(16,1024,1024,4) * (16,1,1,4)          [lds]", (16, 1024, 1024, 4), dense((16, 1024, 1024, 4)), (4, 0, 0, 1), op=2)
case("A.T + B (8192,8192)                    [tile]", (M, M), (1, M), (M, 1))
case("A.T + B.T (8192,8192)                  [tile]", (M, M), (1, M), (1, M))
case("(32,64,512,128) perm(0,2,1,3) + dense  [row]", (32, 512, 64, 128), (64 * 512 * 128, 128, 512 * 128, 1), dense((32, 512, 64, 128)))
case("(64,1024,1024) perm(0,2,1) + dense     [tile]", (64, 1024, 1024), (1024 * 1024, 1, 1024), dense((64, 1024, 1024)))
case("(1024,256,256) perm(2,1,0) + dense     [tile]", (256, 256, 1024), (1, 256, 65536), dense((256, 256, 1024)))
case("A[1:-1,1:-1] + B[1:-1,1:-1] (8190,8190)[row, unaligned]", (M - 2, M - 2), (M, 1), (M, 1), off_a=M + 1, off_b=M + 1)
case("A[:, ::2] + B[:, ::2] (8192,4096)      [gather]", (M, M // 2), (M, 2), (M, 2))
case("A[::2, :] + B[::2, :] (4096,8192)      [row]", (M // 2, M), (2 * M, 1), (2 * M, 1))
case("column A[:,5] + B[:,7] (8192)          [gather]", (M,), (M,), (M,), off_a=5, off_b=7)
case("(2^24,3) / (2^24,1) per-pixel scale     [short rows]", (1 << 24, 3), (3, 1), (1, 0), op=3)
case("f64 (8192,4096) * (1,4096)             [flat]", (M, 4096), (4096, 1), (0, 1), op=2, dt=np.float64)
case("f64 A.T + B (4096,4096)                [tile]", (4096, 4096), (1, 4096), (4096, 1), dt=np.float64)
case("i32 (8192,8192) + (1,8192)             [flat]", (M, M), (M, 1), (0, 1), dt=np.int32)
case("pow (4096,4096) ^ (1,4096)             [flat]", (4096, 4096), (4096, 1), (0, 1), op=4)

DT = {np.float32: 0, np.float64: 1, np.int32: 2, np.int64: 3}
print("%-62s %11s %9s %8s %7s" % ("case", "n", "us", "GB/s", "% peak"))
for name, shape, sa, sb, off_a, off_b, op, dt in cases:
    n = int(np.prod(shape)); esz = np.dtype(dt).itemsize
    na, nb = off_a + span(shape, sa), off_b + span(shape, sb)
    if dt == np.float32:
        a = lib.uniform_f32(na, 1, 0.5, 2.0); b = lib.uniform_f32(nb, 2, 0.5, 2.0)
    else:
        a = lib.empty((na,), dt); b = lib.empty((nb,), dt)
        lib.upload(a.ptr, (np.arange(na) % 251 + 1).astype(dt)); lib.upload(b.ptr, (np.arange(nb) % 13 + 1).astype(dt))
    out = lib.empty((n,), dt)
    args = (C.c_int(op), C.c_int(DT[dt]), C.c_void_p(a.ptr + off_a * esz), i64(sa), C.c_void_p(b.ptr + off_b * esz), i64(sb), i64(shape), C.c_int(len(shape)), C.c_void_p(out.ptr))
    t = timeit(lib.c.smhip_elementwise, args, 50)
    byts = esz * (distinct(shape, sa) + distinct(shape, sb) + n)
    print("%-62s %11d %9.1f %8.0f %6.1f%%" % (name, n, t, byts / t * 1e-3, byts / t * 1e-3 / 80.0), flush=True)
    del a, b, out
