"""Image-layout changes as broadcast problems: out = permuted view + dense, f32.  NHWC -> NCHW is out (B, C, HW) = x(B, HW, C).T-per-plane,
NCHW -> NHWC the reverse; small C takes the record kernel (SMHIP_RECORD_KERNEL=0: the kernels before it), C >= 16 one of the tile kernel's patches.
    python tools/layout_rates.py"""
import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
def timeit(fn, steps=20):
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    res = []
    for _ in range(3):
        lib.synchronize(); lib.record(e0)
        for _ in range(steps): fn()
        lib.record(e1); lib.synchronize()
        res.append(lib.elapsed_ms(e0, e1) / steps * 1000)
    return sorted(res)[1]
print("SMHIP_RECORD_KERNEL =", os.environ.get("SMHIP_RECORD_KERNEL", "(on)"))
f32 = C.c_int(0)
for B, HW, Cc in ((256, 224 * 224, 3), (64, 512 * 512, 3), (256, 224 * 224, 4), (512, 112 * 112, 8), (256, 56 * 56, 64), (64, 56 * 56, 256), (16, 1024 * 1024, 3), (1024, 64 * 64, 3)):
    n = B * HW * Cc
    x = lib.uniform_f32(n, 1, -1, 1); y = lib.uniform_f32(n, 2, -1, 1); out = lib.empty((n,), np.float32)
    # NHWC -> NCHW: out (B, C, HW); x is (B, HW, C): strides (HW*C, 1, C) over (B, C, HW); y dense
    to_nchw = lambda: lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(x.ptr), i64([HW * Cc, 1, Cc]), C.c_void_p(y.ptr), i64([HW * Cc, HW, 1]), i64([B, Cc, HW]), C.c_int(3), C.c_void_p(out.ptr))
    # NCHW -> NHWC: out (B, HW, C); x is (B, C, HW): strides (HW*C, 1, HW) over (B, HW, C)
    to_nhwc = lambda: lib.c.smhip_elementwise(C.c_int(0), f32, C.c_void_p(x.ptr), i64([HW * Cc, 1, HW]), C.c_void_p(y.ptr), i64([HW * Cc, Cc, 1]), i64([B, HW, Cc]), C.c_int(3), C.c_void_p(out.ptr))
    t1, t2 = timeit(to_nchw), timeit(to_nhwc)
    print("B %4d  HW %8d  C %3d  %5.0f MiB/operand   NHWC->NCHW %8.1f us %5.1f %%   NCHW->NHWC %8.1f us %5.1f %%" % (B, HW, Cc, n * 4 / 2**20, t1, 12.0 * n / t1 * 1e-3 / 80, t2, 12.0 * n / t2 * 1e-3 / 80), flush=True)
    del x, y, out; lib.pool_trim()
