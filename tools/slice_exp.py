"""Row kernel on sliced 2-D views: which of (row extent, start offset, pitch) costs the 15 % seen for A[1:-1,1:-1]?"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
M = 8192
a = lib.uniform_f32(M * M + 64, 1, 0.5, 2.0); b = lib.uniform_f32(M * M + 64, 2, 0.5, 2.0); out = lib.empty((M * M,), np.float32)
def run(name, rows, cols, off, pitch=M, out_off=0):
    shape, st = (rows, cols), (pitch, 1)
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a.ptr + off * 4), i64(st), C.c_void_p(b.ptr + off * 4), i64(st), i64(shape), C.c_int(2), C.c_void_p(out.ptr + out_off * 4))
    fn = lib.c.smhip_elementwise
    for _ in range(5): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(50): fn(*args)
    lib.record(e1); lib.synchronize()
    t = lib.elapsed_ms(e0, e1) / 50 * 1000
    print("%-44s %8.1f us %7.0f GB/s" % (name, t, 12.0 * rows * cols / t * 1e-3), flush=True)
run("full 8192 x 8192 (contiguous kernel)", M, M, 0)
run("A[:, :8188]  aligned start, cols % 4 == 0", M, 8188, 0)
run("A[:, :8160]  cols % 32 == 0 (128 B)", M, 8160, 0)
run("A[:, :8190]  cols % 4 == 2", M, 8190, 0)
run("A[:, 1:8189] start + 1, cols 8188", M, 8188, 1)
run("A[:, 4:]     start + 4, cols 8188", M, 8188, 4)
run("A[1:-1,1:-1] 8190 x 8190", M - 2, M - 2, M + 1)
run("A[:, :4096]  half rows", M, 4096, 0)
run("A[:, :8188] pitch 8188 (dense, as 2-D) ", M, 8188, 0, pitch=8188)
