"""Operator chains on resident arrays: every launch reads what the previous one wrote (ping-pong), next to the bench.py
setting where every launch re-reads the same operands.  f32; sizes are per array."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fns, steps=60):
    k = len(fns)
    for i in range(10): fns[i % k]()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for i in range(steps): fns[i % k]()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize()
del x
f32 = C.c_int(0)
one = C.c_float(1.0000001)
print("%-44s %8s %10s %10s" % ("case", "MiB", "same", "chain"))
def row(name, mib, byts, same, chain): print("%-44s %8d %7.1f us %5.1f%% %7.1f us %5.1f%%" % (name, mib, same, byts / same * 1e-3 / 80, chain, byts / chain * 1e-3 / 80), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "rotate":
    # cold operands: launches walk K different (a, out) pairs, K x 2 x size >= 2 GiB, so nothing a launch reads or writes
    # was touched recently -- the setting the read / write policies are NOT tuned for
    print("%-44s %8s %10s" % ("case (rotating operands)", "MiB", "rotate"))
    for mib in (32, 64, 128):
        n = mib << 18
        K = max(2, 1024 // mib)
        srcs = [lib.uniform_f32(n, 10 + k, 0.5, 2.0) for k in range(K)]; dsts = [lib.empty((n,), np.float32) for _ in range(K)]
        fns = [(lambda s_, d_: (lambda: lib.c.smhip_array_scalar(C.c_int(2), f32, C.c_void_p(s_.ptr), C.byref(one), C.c_size_t(n), C.c_void_p(d_.ptr))))(srcs[k], dsts[k]) for k in range(K)]
        t = timeit(fns, steps=3 * K)
        print("%-44s %8d %7.1f us %5.1f%%" % ("a * s (array_scalar), K = %d pairs" % K, mib, t, 8.0 * n / t * 1e-3 / 80), flush=True)
        del srcs, dsts, fns; lib.pool_trim()
    sys.exit(0)
for mib in (16, 32, 64, 128, 256):
    n = mib << 18
    a = lib.uniform_f32(n, 1, 0.5, 2.0); b = lib.uniform_f32(n, 2, 0.5, 2.0); c = lib.empty((n,), np.float32)
    def sc(src, dst, op=2): return lambda: lib.c.smhip_array_scalar(C.c_int(op), f32, C.c_void_p(src.ptr), C.byref(one), C.c_size_t(n), C.c_void_p(dst.ptr))
    row("a * s (array_scalar)", mib, 8.0 * n, timeit([sc(a, c)]), timeit([sc(a, c), sc(c, a)]))
    def ad(x, y, dst): return lambda: lib.c.smhip_contiguous(C.c_int(0), f32, C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_void_p(dst.ptr), C.c_size_t(n))
    row("a + b (contiguous)", mib, 12.0 * n, timeit([ad(a, b, c)]), timeit([ad(a, b, c), ad(c, b, a)]))
    if mib <= 128:
        row("pow(a, 1.0000001) (array_scalar)", mib, 8.0 * n, timeit([sc(a, c, 4)]), timeit([sc(a, c, 4), sc(c, a, 4)]))
    cols = 4096; rows = n // cols
    r = lib.uniform_f32(cols, 3, 0.99, 1.01)
    def rw(src, dst): return lambda: lib.c.smhip_elementwise(C.c_int(2), f32, C.c_void_p(src.ptr), i64([cols, 1]), C.c_void_p(r.ptr), i64([0, 1]), i64([rows, cols]), C.c_int(2), C.c_void_p(dst.ptr))
    row("(R,4096) * (1,4096) (row kernel)", mib, 8.0 * n, timeit([rw(a, c)]), timeit([rw(a, c), rw(c, a)]))
    del a, b, c, r
    lib.pool_trim()
