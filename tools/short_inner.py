"""Dense rows with a short inner extent (17 ... 100 elements, not multiples of the vector width) against a row, a column, a dense partner: f32.   python tools/short_inner.py"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
F64 = len(sys.argv) > 1 and sys.argv[1] == "f64"
DTC, ESZ = (1, 8) if F64 else (0, 4)
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
for c in (2, 3, 5, 13, 17, 20, 31, 33, 50, 63, 64, 65, 100, 127, 250, 500, 1000, 2000):
    r = (1 << (25 if F64 else 26)) // c
    n = r * c
    x = lib.uniform_f32(n * ESZ // 4, 1, 1, 2); y = lib.uniform_f32(max(n, 4096) * ESZ // 4, 2, 1, 2); out = lib.empty((n * ESZ // 4,), np.float32)
    line = "rows of %3d:" % c
    for name, ys, alg in (("+row", (0, 1), 2.0 * ESZ), ("+col", (1, 0), 2.0 * ESZ), ("+dense view (pitch c+3)", None, 3.0 * ESZ), ("view*2", "scalar", 2.0 * ESZ)):
        if ys == "scalar":
            rr = (1 << (25 if F64 else 26)) // (c + 3)
            fn = lambda: lib.c.smhip_elementwise(C.c_int(2), C.c_int(DTC), C.c_void_p(x.ptr), i64([c + 3, 1]), C.c_void_p(y.ptr), i64([0, 0]), i64([rr - 1, c]), C.c_int(2), C.c_void_p(out.ptr))
            m = (rr - 1) * c
        elif ys is None:
            # x as a view of rows with a pitch of c + 3 (a column slice of a wider array) plus dense y
            rr = (1 << (25 if F64 else 26)) // (c + 3)
            fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(DTC), C.c_void_p(x.ptr), i64([c + 3, 1]), C.c_void_p(y.ptr), i64([c, 1]), i64([rr - 1, c]), C.c_int(2), C.c_void_p(out.ptr))
            m = (rr - 1) * c
        else:
            fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(DTC), C.c_void_p(x.ptr), i64([c, 1]), C.c_void_p(y.ptr), i64(ys), i64([r, c]), C.c_int(2), C.c_void_p(out.ptr))
            m = n
        t = timeit(fn)
        line += "  %s %5.1f %%" % (name, alg * m / t * 1e-3 / 80)
    print(line, flush=True)
    del x, y, out; lib.pool_trim()
