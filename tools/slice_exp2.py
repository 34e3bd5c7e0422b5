"""Row kernel where only ONE operand has a padded pitch: is the 15 % the kernel or the access pattern?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
M = 8192
a = lib.uniform_f32(M * (M + 64), 1, 0.5, 2.0); b = lib.uniform_f32(M * (M + 64), 2, 0.5, 2.0); out = lib.empty((M * M,), np.float32)
def run(name, rows, cols, pa, pb):
    shape = (rows, cols)
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), i64((pa, 1)), C.c_void_p(b.ptr), i64((pb, 1)), i64(shape), C.c_int(2), C.c_void_p(out.ptr))
    fn = lib.c.smhip_elementwise
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(50): fn(*args)
    lib.record(e1); lib.synchronize()
    t = lib.elapsed_ms(e0, e1) / 50 * 1000
    print("%-60s %8.1f us %7.0f GB/s" % (name, t, 12.0 * rows * cols / t * 1e-3), flush=True)
run("dense (contiguous kernel)", M, M, M, M)
run("a pitch 8192, b pitch 8196 (b drifts 16 B/row)", M, M, M, M + 4)
run("a pitch 8192, b pitch 8224 (b drifts 128 B/row)", M, M, M, M + 32)
run("a, b pitch 8196 (out drifts)", M, M, M + 4, M + 4)
run("a, b pitch 8224 (out drifts 128 B/row)", M, M, M + 32, M + 32)
run("a, b pitch 8256 (out drifts 256 B/row)", M, M, M + 64, M + 64)
run("a pitch 8196, b pitch 8224", M, M, M + 4, M + 32)
