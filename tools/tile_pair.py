"""out(N, N) = A.T + B at N = 12288 (rows on 128-byte lines) and N = 12287 (rows off them): ten launches each, for the counter
passes of tools/pmc_tile_odd.sh and for the rates.   python tools/tile_pair.py [N_aligned N_ragged]"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
def i64(seq): return (C.c_int64 * len(seq))(*[int(s) for s in seq])
sizes = [int(x) for x in sys.argv[1:3]] if len(sys.argv) > 2 else [12288, 12287]
for N in sizes:
    A = lib.uniform_f32(N * N, 1, -1, 1); B = lib.uniform_f32(N * N, 2, -1, 1); out = lib.empty((N * N,), np.float32)
    fn = lambda: lib.c.smhip_elementwise(C.c_int(0), C.c_int(0), C.c_void_p(A.ptr), i64([1, N]), C.c_void_p(B.ptr), i64([N, 1]), i64([N, N]), C.c_int(2), C.c_void_p(out.ptr))
    for _ in range(3): fn()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(10): fn()
    lib.record(e1); lib.synchronize()
    t = lib.elapsed_ms(e0, e1) / 10 * 1000
    print("A.T + B  %5d x %5d  %9.1f us  %5.1f %%" % (N, N, t, 12.0 * N * N / t * 1e-3 / 80), flush=True)
    del A, B, out; lib.pool_trim()
