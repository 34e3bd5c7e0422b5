"""out(P, Q) = A.T + B with A of shape (Q, P): the LDS tile kernel over rectangular shapes, replayed and (where four operand sets fit in 40 GiB) cold.
python tools/tile_shapes.py [library|-] [fine|f64|skinny|tiny]   -- tools/tile_variants.sh builds libraries that differ in the patch walk and q extent."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
import os
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "-" else sma.load()
print("library:", sys.argv[1] if len(sys.argv) > 1 else "built", " SMHIP_TILE_QB =", os.environ.get("SMHIP_TILE_QB", os.environ.get("SMHIP_TILE_WIDE", "(auto)")), flush=True)
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
f32 = C.c_int(0)
F64 = len(sys.argv) > 2 and sys.argv[2] == "f64"   # doubles: 64 x 64 (512 B) and 64 x 128 (1024 B) patches
DT, ESZ = (C.c_int(1), 8) if F64 else (f32, 4)
SHAPES = ((4096, 4096), (2048, 32768), (32768, 2048), (4096, 16384), (16384, 4096), (8192, 8192), (6144, 16384), (16384, 6144), (10240, 10240),
          (8192, 16384), (16384, 8192), (12288, 12288), (4096, 65536), (65536, 4096), (16384, 16384), (8192, 65536), (65536, 8192))
if F64:
    SHAPES = ((4096, 4096), (4096, 8192), (8192, 4096), (6144, 6144), (8192, 8192), (4096, 32768), (32768, 4096), (12288, 12288))
if len(sys.argv) > 2 and sys.argv[2] == "skinny":   # one plane extent far below a patch row
    SHAPES = ((4194304, 32), (32, 4194304), (2097152, 64), (1048576, 128), (128, 1048576), (524288, 256), (256, 524288), (262144, 512), (512, 262144), (2097152, 16), (16, 2097152))
if len(sys.argv) > 2 and sys.argv[2] == "tiny":   # plane extents below the tile kernel's 16: AoS <-> SoA of small records (gather / strided kernels)
    SHAPES = ((3, 16777216), (16777216, 3), (4, 16777216), (16777216, 4), (8, 8388608), (8388608, 8), (12, 4194304), (4194304, 12), (16, 4194304), (4194304, 16), (24, 4194304), (4194304, 24))
if len(sys.argv) > 2 and sys.argv[2] == "pow2":   # the record kernel: long extents that are and are not a power of two
    SHAPES = ((8, 8388608), (8, 8388608 + 4096), (16, 4194304), (16, 4194304 + 4096), (32, 2097152), (32, 2097152 + 4096), (4, 16777216), (4, 16777216 + 4096),
              (8388608, 8), (8388608 + 4096, 8), (6, 8388608), (24, 2097152))
if len(sys.argv) > 2 and sys.argv[2] == "odd":   # extents that are not a multiple of the vector width: the tile kernel's element form
    SHAPES = ((8192, 8192), (8191, 8191), (8190, 8190), (8188, 8188), (8191, 8192), (8192, 8191), (12287, 12287), (12284, 12284), (16383, 16383))
if len(sys.argv) > 2 and sys.argv[2] == "mid":   # a small extent between the record kernel's 32 and a whole patch
    SHAPES = ((36, 1 << 21), (40, 1 << 21), (48, 1 << 21), (56, 1 << 21), (64, 1 << 21), (72, 1 << 20), (96, 1 << 20), (1 << 21, 36), (1 << 21, 40), (1 << 21, 48), (1 << 21, 72), (1 << 20, 100))
if len(sys.argv) > 2 and sys.argv[2] == "fine":   # around the size where the wide patch takes over
    SHAPES = ((6144, 8192), (8192, 6144), (8192, 8192), (8192, 8704), (8704, 8192), (9216, 9216), (8192, 10240), (10240, 8192), (9728, 9728), (4096, 20480), (20480, 4096),
              (10240, 10240), (8192, 12288), (12288, 8192), (11264, 11264), (2048, 65536), (65536, 2048))
for P, Q in SHAPES:
    n = P * Q
    nsets = 4 if 12 * n * ESZ <= 40 << 30 else 1
    if F64:
        sets = [(lib.full((n,), 1.5 + k, np.float64), lib.full((n,), 2.5 + k, np.float64), lib.empty((n,), np.float64)) for k in range(nsets)]
    else:
        sets = [(lib.uniform_f32(n, 1 + k, -1, 1), lib.uniform_f32(n, 9 + k, -1, 1), lib.empty((n,), np.float32)) for k in range(nsets)]
    state = [0]
    def one(rotate):
        A, B, out = sets[state[0] % nsets if rotate else 0]; state[0] += 1
        lib.c.smhip_elementwise(C.c_int(0), DT, C.c_void_p(A.ptr), i64([1, P]), C.c_void_p(B.ptr), i64([Q, 1]), i64([P, Q]), C.c_int(2), C.c_void_p(out.ptr))
    t = timeit(lambda: one(False))
    tc = timeit(lambda: one(True), 12) if nsets > 1 else float("nan")
    print("out %5d x %5d  %6.0f MiB/operand  replay %8.1f us %5.1f %%   cold %8.1f us %5.1f %%" % (P, Q, n * ESZ / 2**20, t, 3.0 * ESZ * n / t * 1e-3 / 80, tc, 3.0 * ESZ * n / tc * 1e-3 / 80), flush=True)
    del sets; lib.pool_trim()
