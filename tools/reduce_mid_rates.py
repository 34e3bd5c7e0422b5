"""Small and mid-size reductions, 2^16 .. 2^26 elements: f32 / f64 sum, dot, fused add+sum and the complex<double> dot --
the kernels alone (async entry points, HIP events) and the synchronous call with its scalar read-back (host clock).
Percentages are of 8 TB/s on the algorithmic bytes.   python tools/reduce_mid_rates.py [lib.so]
SMHIP_REDUCE_ONE_LAUNCH=0 shows the two-launch form at every size; SMHIP_REDUCE_ONE_TILES=<t> moves the one-launch limit."""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()
c = lib.c

def events(fn, steps):
    for _ in range(10): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000

def host(fn, steps):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    return (time.perf_counter() - t0) / steps * 1e6

nmax = 1 << 26
bufs = {}
for dt in (np.float32, np.float64):
    a, b, o = lib.empty((nmax,), dt), lib.empty((nmax,), dt), lib.empty((nmax,), dt)
    one = np.array([1], dtype=dt)
    for x in (a, b): c.smhip_fill(C.c_int(sma.DTYPES[np.dtype(dt)]), C.c_void_p(x.ptr), one.ctypes.data_as(C.c_void_p), C.c_size_t(nmax))
    bufs[dt] = (a, b, o)
sp = lib.alloc(16)
hres = (C.c_double * 2)()
print("%-26s %6s | %9s %7s | %9s %7s" % ("reduction", "log2 n", "async us", "% peak", "sync us", "% peak"))
for lg in (16, 18, 20, 21, 22, 23, 24, 26):
    n = 1 << lg
    steps = 400 if lg <= 22 else 100
    for dt in (np.float32, np.float64):
        a, b, o = bufs[dt]
        code, esz = C.c_int(sma.DTYPES[np.dtype(dt)]), np.dtype(dt).itemsize
        pa, pb, po, psp, cn = C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(o.ptr), C.c_void_p(sp), C.c_size_t(n)
        rows = (("sum", lambda: c.smhip_sum_async(code, pa, cn, psp), lambda: c.smhip_sum(code, pa, cn, hres), esz * n),
                ("dot", lambda: c.smhip_dot_async(code, pa, pb, cn, psp), lambda: c.smhip_dot(code, pa, pb, cn, hres), 2 * esz * n),
                ("fused add+sum", lambda: c.smhip_contiguous_sum_async(C.c_int(0), code, pa, pb, po, cn, psp), None, 3 * esz * n))
        for name, fa, fs, byts in rows:
            ta = events(fa, steps)
            ts = host(fs, steps // 4) if fs else float("nan")
            print("%-26s %6d | %9.2f %6.1f%% | %9.2f %6.1f%%" % ("%s %s" % (np.dtype(dt).name, name), lg, ta, byts / ta * 1e-3 / 80, ts, byts / ts * 1e-3 / 80 if fs else float("nan")), flush=True)
    if lg <= 25:  # complex<double>: n pairs = 2 n doubles per operand
        a, b, _ = bufs[np.float64]
        pa, pb, cn = C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_size_t(n)
        ta = events(lambda: c.smhip_dot_c64_async(pa, pb, cn, C.c_void_p(sp)), steps)
        ts = host(lambda: c.smhip_dot_c64(pa, pb, cn, hres), steps // 4)
        print("%-26s %6d | %9.2f %6.1f%% | %9.2f %6.1f%%" % ("complex128 dot", lg, ta, 32.0 * n / ta * 1e-3 / 80, ts, 32.0 * n / ts * 1e-3 / 80), flush=True)
lib.free(sp)
