"""Mid-size launches of the library's own kernels on COLD operands (VERDICT r02 "next" #2).

Three settings per case, f32, sizes per array:
  same    every launch re-reads the same operands (bench.py's replay setting: the Infinity Cache feeds it)
  rotate  launches walk K disjoint operand sets, K x footprint >= 2.5 GiB: nothing a launch touches was touched recently
  chain   ping-pong, each launch reads what the previous one wrote
Cases: a*s (smhip_array_scalar, 1R+1W), a+b (smhip_contiguous, 2R+1W), (R,4096)*(1,4096) (smhip_elementwise: config 3's shape
at 64 MiB).

    python tools/cold_rates.py                       # the table
    python tools/cold_rates.py --pmc scalar:64       # ONE case for a rocprofv3 --pmc pass: 30 launches `same`, then K warm +
                                                     # 30 launches `rotate` (tools/pmc_cold.sh splits the rows by order)
"""
import argparse
import ctypes as C
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma

ap = argparse.ArgumentParser()
ap.add_argument("--pmc", default=None, help="case:MiB, e.g. scalar:64, add:16, row:64")
ap.add_argument("--sizes", default="16,32,64,128")
ap.add_argument("--lib", default=None)
args = ap.parse_args()
lib = sma.load(args.lib) if args.lib else sma.load()
f32 = C.c_int(0)
one = C.c_float(1.0000001)


def i64(seq):
    return (C.c_int64 * len(seq))(*[int(s) for s in seq])


SLAB = 6 << 30
slab = lib.alloc(SLAB)
lib.c.smhip_fill_uniform_f32(C.c_void_p(slab), SLAB // 4, 11, 0, 0.5, 2.0)
row_ptr = lib.uniform_f32(4096, 3, 0.99, 1.01)
lib.synchronize()


def make(case, mib):
    """(launch(set index k, swapped), arrays per set, bytes per launch)"""
    n = mib << 18
    arrays = 3 if case == "add" else 2
    set_bytes = arrays * n * 4
    K = max(2, min(SLAB // set_bytes, (2560 << 20) // set_bytes + 1))

    def ptrs(k):
        base = slab + (k % K) * set_bytes
        return [base + j * n * 4 for j in range(arrays)]

    if case == "scalar":
        def launch(k, swap=False):
            a, o = ptrs(k)
            if swap:
                a, o = o, a
            lib.c.smhip_array_scalar(C.c_int(2), f32, C.c_void_p(a), C.byref(one), C.c_size_t(n), C.c_void_p(o))
        byts = 8.0 * n
    elif case == "add":
        def launch(k, swap=False):
            a, b, o = ptrs(k)
            if swap:
                a, o = o, a
            lib.c.smhip_contiguous(C.c_int(0), f32, C.c_void_p(a), C.c_void_p(b), C.c_void_p(o), C.c_size_t(n))
        byts = 12.0 * n
    else:
        cols = 4096
        rows = n // cols
        sa, sb, shp = i64([cols, 1]), i64([0, 1]), i64([rows, cols])

        def launch(k, swap=False):
            a, o = ptrs(k)
            if swap:
                a, o = o, a
            lib.c.smhip_elementwise(C.c_int(2), f32, C.c_void_p(a), sa, C.c_void_p(row_ptr.ptr), sb, shp, C.c_int(2), C.c_void_p(o))
        byts = 8.0 * n + 4 * cols
    return launch, K, byts


def timeit(fn, steps=60, warm=24):
    for i in range(warm):
        fn(i)
    e0, e1 = lib.event(), lib.event()
    res = []
    for _ in range(5):
        lib.synchronize()
        lib.record(e0)
        for i in range(steps):
            fn(warm + i)
        lib.record(e1)
        lib.synchronize()
        res.append(lib.elapsed_ms(e0, e1) / steps * 1000)
    return sorted(res)[2]


# leave the idle clocks
launch, K, _ = make("add", 64)
for i in range(200 if args.pmc else 2000):
    launch(i)
lib.synchronize()

if args.pmc:
    case, mib = args.pmc.split(":")
    launch, K, byts = make(case, int(mib))
    for i in range(30):
        launch(0)
    lib.synchronize()
    for i in range(K + 30):
        launch(1 + i)
    lib.synchronize()
    print("pmc case %s %s MiB: 30 launches same, then %d + 30 rotate; %.0f bytes per launch" % (case, mib, K, byts))
    sys.exit(0)

print("%-28s %6s %4s | %9s %6s | %9s %6s | %9s %6s" % ("case", "MiB", "K", "same us", "%", "rotate us", "%", "chain us", "%"))
for case in ("scalar", "add", "row"):
    for mib in [int(s) for s in args.sizes.split(",")]:
        launch, K, byts = make(case, mib)
        pct = lambda us: byts / us * 1e-3 / 80
        t_same = timeit(lambda i: launch(0))
        t_rot = timeit(lambda i: launch(i))
        t_chain = timeit(lambda i: launch(0, swap=bool(i & 1)))
        name = {"scalar": "a * s      (1R+1W)", "add": "a + b      (2R+1W)", "row": "(R,4096)*(1,4096)"}[case]
        print("%-28s %6d %4d | %9.2f %5.1f%% | %9.2f %5.1f%% | %9.2f %5.1f%%" % (name, mib, K, t_same, pct(t_same), t_rot, pct(t_rot), t_chain, pct(t_chain)), flush=True)
