"""Few long rows against one row, f32: (rows, cols) + (1, cols) through the flat tile kernel's column-block walk.  SMHIP_ROWS_WALK_LOG2 = -1
switches the walk off, k makes 2^k consecutive workgroups visit one row before the next row's same columns.     python tools/rows_walk.py"""
import os, sys, ctypes as C
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
out_line = ["walk log2 = %3s |" % os.environ.get("SMHIP_ROWS_WALK_LOG2", "def")]
for rows, cols in ((8, 1 << 23), (4, 1 << 24), (16, 1 << 22), (64, 1 << 20), (2, 1 << 25), (256, 1 << 20), (3, 3 << 22)):
    n = rows * cols
    x = lib.uniform_f32(n, 1, -1, 1); y = lib.uniform_f32(cols, 2, -1, 1); out = lib.empty((n,), np.float32)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(0), C.c_void_p(x.ptr), i64([cols, 1]), C.c_void_p(y.ptr), i64([0, 1]), i64([rows, cols]), C.c_int(2), C.c_void_p(out.ptr))
    t = timeit(fn)
    out_line.append(" (%d,2^%.1f) %5.1f %%" % (rows, np.log2(cols), 4.0 * (2 * n + cols) / t * 1e-3 / 80))
    del x, y, out; lib.pool_trim()
print("".join(out_line), flush=True)
