"""One operator and a synchronisation, n = 25 ... 8192 f32: the latency of a single (recorded or launched) operator, and the
throughput of 24 independent ones before one synchronisation.   SMHIP_TINY_BATCH=0 python tools/tiny_latency.py  for one launch each."""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load()
for n in (25, 256, 1024, 2048, 4096, 8192):
    a = lib.to_device(np.ones(n, dtype=np.float32)); b = lib.to_device(np.ones(n, dtype=np.float32))
    outs = [lib.empty((n,), np.float32) for _ in range(24)]
    lib.synchronize()
    c = lib.c
    args = [(C.c_int(0), C.c_int(0), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(o.ptr), C.c_size_t(n)) for o in outs]
    def one():
        c.smhip_contiguous(*args[0]); c.smhip_synchronize()
    def many():
        for x in args: c.smhip_contiguous(*x)
        c.smhip_synchronize()
    for f in (one, many): [f() for _ in range(50)]
    t0 = time.perf_counter(); [one() for _ in range(500)]; t1 = (time.perf_counter() - t0) / 500 * 1e6
    t0 = time.perf_counter(); [many() for _ in range(200)]; t24 = (time.perf_counter() - t0) / 200 * 1e6
    print("n = %5d   one operator + synchronize %7.2f us    24 independent + synchronize %7.2f us (%.2f us each)" % (n, t1, t24, t24 / 24), flush=True)
