"""Which side's misalignment costs the ragged transposes their rate?  out = A.T + B through the tile kernel (f32, N ~ 12288: past the
Infinity Cache), with (1) rows on 128-byte lines and the BASE pointers moved off them by k elements, per operand, and (2) rows off the
lines (N = 12287) with one operand at a time given an aligned pitch of 12288.   python tools/tile_align_probe.py"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
M = 12288
A = lib.uniform_f32(M * M + 64, 1, -1, 1); B = lib.uniform_f32(M * M + 64, 2, -1, 1); out = lib.empty((M * M + 64,), np.float32)
def run(label, n, a_off, a_pitch, b_off, b_pitch, o_off):
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(0), C.c_void_p(A.ptr + 4 * a_off), i64([1, a_pitch]), C.c_void_p(B.ptr + 4 * b_off), i64([b_pitch, 1]),
                                         i64([n, n]), C.c_int(2), C.c_void_p(out.ptr + 4 * o_off))
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(10): fn()
    lib.record(e1); lib.synchronize()
    t = lib.elapsed_ms(e0, e1) / 10 * 1000
    print("%-64s %9.1f us  %5.1f %%" % (label, t, 12.0 * n * n / t * 1e-3 / 80), flush=True)
run("12288^2, everything on lines", M, 0, M, 0, M, 0)
for k in (1, 8, 16):
    run("12288^2, A (turned) base + %d elements" % k, M, k, M, 0, M, 0)
    run("12288^2, B (direct) base + %d" % k, M, 0, M, k, M, 0)
    run("12288^2, out base + %d" % k, M, 0, M, 0, M, k)
    run("12288^2, B and out base + %d" % k, M, 0, M, k, M, k)
    run("12288^2, all three + %d" % k, M, k, M, k, M, k)
n = M - 1
run("12287^2, dense (all rows off the lines)", n, 0, n, 0, n, 0)
run("12287^2, A pitch 12288 (turned side on lines)", n, 0, M, 0, n, 0)
run("12287^2, B pitch 12288 (direct side on lines, out off)", n, 0, n, 0, M, 0)
run("12287^2, A and B pitch 12288 (only out off the lines)", n, 0, M, 0, M, 0)
