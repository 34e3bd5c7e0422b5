"""Rates of the remaining entry points: fused 3-operand, strided copy / assignment, dense copy of views (LEFT), repeat."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=30):
    for _ in range(5): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
def show(name, t, byts): print("%-58s %9.1f us %7.0f GB/s %6.1f%%" % (name, t, byts / t * 1e-3, byts / t * 1e-3 / 80), flush=True)
M = 8192
n = M * M
a = lib.uniform_f32(n, 1, 0.5, 2.0); b = lib.uniform_f32(n, 2, 0.5, 2.0); c = lib.uniform_f32(n, 3, 0.5, 2.0); out = lib.empty((n,), np.float32)
f32 = C.c_int(0)
show("fused (a + b) * c, arrays, 2^26", timeit(lambda: lib.c.smhip_fused_contiguous(C.c_int(0), C.c_int(2), f32, C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(c.ptr), None, C.c_void_p(out.ptr), C.c_size_t(n))), 16 * n)
two = np.array([2.0], dtype=np.float32)
show("fused (a + b) * 2, scalar, 2^26", timeit(lambda: lib.c.smhip_fused_contiguous(C.c_int(0), C.c_int(2), f32, C.c_void_p(a.ptr), C.c_void_p(b.ptr), None, two.ctypes.data_as(C.c_void_p), C.c_void_p(out.ptr), C.c_size_t(n))), 12 * n)
def copy(ss, sd, shape, off_s=0, off_d=0): return lambda: lib.c.smhip_copy_strided(f32, C.c_void_p(a.ptr + 4 * off_s), i64(ss), C.c_void_p(out.ptr + 4 * off_d), i64(sd), i64(shape), C.c_int(len(shape)))
show("copy dense -> dense (memcpy)", timeit(copy((M, 1), (M, 1), (M, M))), 8 * n)
show("assign A[1:-1,1:-1] = B[1:-1,1:-1]", timeit(copy((M, 1), (M, 1), (M - 2, M - 2), M + 1, M + 1)), 8 * (M - 2) * (M - 2))
show("assign dense <- transposed (strided read)", timeit(copy((1, M), (M, 1), (M, M))), 8 * n)
show("assign transposed <- dense (strided write)", timeit(copy((M, 1), (1, M), (M, M))), 8 * n)
show("assign A[:, ::2] = B[:, ::2]", timeit(copy((M, 2), (M, 2), (M, M // 2))), 8 * n // 2)
zeros = (0, 0)
def left(sa, shape): return lambda: lib.c.smhip_elementwise(C.c_int(5), f32, C.c_void_p(a.ptr), i64(sa), C.c_void_p(a.ptr), i64([0] * len(shape)), i64(shape), C.c_int(len(shape)), C.c_void_p(out.ptr))
show("contiguous() of A.T (LEFT through the tile kernel)", timeit(left((1, M), (M, M))), 8 * n)
show("contiguous() of A[:, ::2]", timeit(left((M, 2), (M, M // 2))), 8 * n // 2)
show("repeat(4) of 2^24 elements ((N,4) strides (1,0))", timeit(left((1, 0), (n // 4, 4))), 4 * (n // 4) + 4 * n)
show("repeat(4, axis=0) of (2048,8192) ((2048,4,8192) s (8192,0,1))", timeit(left((M, 0, 1), (2048, 4, M))), 4 * 2048 * M + 4 * n)
# a whole expression in one pass against the operator chain it replaces
d = lib.uniform_f32(n, 4, 0.5, 2.0)
t1 = lib.empty((n,), np.float32); t2 = lib.empty((n,), np.float32)
ptrs = (C.c_void_p * 4)(a.ptr, b.ptr, c.ptr, d.ptr)
te = timeit(lambda: lib.c.smhip_fused_expr(b"(a0 + a1) * a2 - a3", f32, ptrs, C.c_int(4), None, C.c_int(0), C.c_void_p(out.ptr), C.c_size_t(n)))
show("expr (a0 + a1) * a2 - a3, one pass, 2^26", te, 20 * n)
def chain():
    lib.c.smhip_contiguous(C.c_int(0), f32, C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(t1.ptr), C.c_size_t(n))
    lib.c.smhip_contiguous(C.c_int(2), f32, C.c_void_p(t1.ptr), C.c_void_p(c.ptr), C.c_void_p(t2.ptr), C.c_size_t(n))
    lib.c.smhip_contiguous(C.c_int(1), f32, C.c_void_p(t2.ptr), C.c_void_p(d.ptr), C.c_void_p(out.ptr), C.c_size_t(n))
tc = timeit(chain)
print("%-58s %9.1f us   (the one-pass form is %.2fx faster)" % ("the same as three operator calls", tc, tc / te))
sp = lib.alloc(8)
ptr2 = (C.c_void_p * 2)(a.ptr, b.ptr)
ts = timeit(lambda: lib.c.smhip_fused_expr_sum_async(b"(a0 - a1) * (a0 - a1)", f32, ptr2, C.c_int(2), None, C.c_int(0), None, C.c_size_t(n), C.c_void_p(sp)))
show("expr_sum (a0 - a1)^2, reduce only, 2^26", ts, 8 * n)
