"""Every built-in Op x element type through smhip_contiguous / smhip_array_scalar at 1 GiB per operand:
achieved HBM rate (algorithmic bytes / HIP-event time).  Evidence table for DESIGN.md."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 else sma.load()
ONLY = sys.argv[2].split(',') if len(sys.argv) > 2 else None  # e.g. f64
def timeit(fn, args, steps=30):
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps
GiB = 1 << 30
# pre-warm clocks
x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(300): lib.c.smhip_array_scalar(C.c_int(4), C.c_int(0), C.c_void_p(x.ptr), C.byref(C.c_float(2.5)), C.c_size_t(1 << 26), C.c_void_p(x.ptr))
lib.synchronize()
print("%-5s %-4s %-14s %10s %9s %7s" % ("dtype", "op", "form", "ms", "GB/s", "% peak"))
for dtn, dt, code in (("f32", np.float32, 0), ("f64", np.float64, 1), ("i32", np.int32, 2), ("i64", np.int64, 3)):
    if ONLY and dtn not in ONLY: continue
    n = GiB // np.dtype(dt).itemsize
    # inputs: positive floats in (0.5, 2) / ints in [1, 1000]: fine for every op incl. div and pow
    if dt in (np.float32, np.float64):
        base = lib.uniform_f32(GiB // 4, 3, 0.5, 2.0)  # bit patterns reused as f64 when dt is f64: still finite positives
        a = sma.DeviceArray(lib, base.base_ptr, dt, (n,), (1,), 0, base._owner)
        b = lib.empty((n,), dt); lib.c.smhip_copy(C.c_void_p(b.ptr), C.c_void_p(a.ptr), C.c_size_t(GiB))
        if dt == np.float64:
            one = np.array([1.5], dtype=dt)
            lib.c.smhip_fill(C.c_int(code), C.c_void_p(a.ptr), one.ctypes.data_as(C.c_void_p), C.c_size_t(n))
            lib.c.smhip_fill(C.c_int(code), C.c_void_p(b.ptr), one.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    else:
        a = lib.full((n,), 7, dt); b = lib.full((n,), 3, dt)
    out = lib.empty((n,), dt)
    for opn, op in (("add", 0), ("sub", 1), ("mul", 2), ("div", 3), ("pow", 4)):
        ms = timeit(lib.c.smhip_contiguous, (C.c_int(op), C.c_int(code), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(out.ptr), C.c_size_t(n)))
        print("%-5s %-4s %-14s %10.4f %9.0f %6.1f%%" % (dtn, opn, "array op array", ms, 3.0 * GiB / ms * 1e-6, 3.0 * GiB / ms * 1e-6 / 80))
    s = np.array([3], dtype=dt)
    for opn, op in (("mul", 2), ("pow", 4)):
        ms = timeit(lib.c.smhip_array_scalar, (C.c_int(op), C.c_int(code), C.c_void_p(a.ptr), s.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_void_p(out.ptr)))
        print("%-5s %-4s %-14s %10.4f %9.0f %6.1f%%" % (dtn, opn, "array op scalar", ms, 2.0 * GiB / ms * 1e-6, 2.0 * GiB / ms * 1e-6 / 80))
    del a, b, out
    lib.pool_trim()
