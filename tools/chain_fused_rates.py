"""smhip_chain against the operator calls it replaces: (A * row + B) * s and relatives, f32, replayed operands ("same":
every step re-reads the same arrays, bench.py's setting) and cold ones ("rotate": steps walk K operand sets, K x footprint
>= 2.5 GiB).  Percentages are of 8 TB/s on the ONE-PASS algorithmic bytes (dense operands + result), so the unfused
column shows what the extra passes cost.
    python tools/chain_fused_rates.py [lib.so]"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
lib = sma.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else sma.load()

def timeit(fns, steps=60):
    k = len(fns)
    for i in range(max(10, k)): fns[i % k]()
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for i in range(steps): fns[i % k]()
    lib.record(e1); lib.synchronize()
    return lib.elapsed_ms(e0, e1) / steps * 1000

def arr2(d, rows, cols): return sma.DeviceArray(lib, d.base_ptr, np.float32, (rows, cols), (cols, 1), 0, d._owner)

x = lib.uniform_f32(1 << 26, 1, 0.5, 2.0)
for _ in range(200): lib.array_scalar(sma.OP_MUL, x, 1.0000001, out=x)
lib.synchronize(); del x
print("%-34s %12s %9s %19s %19s %19s" % ("case", "shape", "B/elem", "chain same", "chain rotate", "operators same"))
half = np.float32(0.5)
SIZES = ((2048, 4096), (4096, 4096), (8192, 4096), (8192, 8192), (16384, 8192))
if len(sys.argv) > 2 and sys.argv[2] == "short": SIZES = ((4096, 4096), (8192, 8192))
for rows, cols in SIZES:
    n = rows * cols
    K = max(2, int(2.5 * 2**30 / (12.0 * n)) + 1)
    sets = []
    for k in range(K):
        A, B = lib.uniform_f32(n, 3 + 10 * k, -1, 1), lib.uniform_f32(n, 4 + 10 * k, -1, 1)
        sets.append((arr2(A, rows, cols), arr2(B, rows, cols), lib.empty((rows, cols), np.float32)))
    row = arr2(lib.uniform_f32(cols, 5, -1, 1), 1, cols)
    col = arr2(lib.uniform_f32(rows, 6, -1, 1), rows, 1)
    t1, t2 = lib.empty((rows, cols), np.float32), lib.empty((rows, cols), np.float32)
    def report(name, bpe, chain_fn, eager_fn):
        calls = [chain_fn(*s_)[0] for s_ in sets]
        same = timeit([calls[0]])
        rot = timeit(calls, steps=4 * K)
        eager = timeit([lambda: eager_fn(*sets[0])])
        byts = bpe * n
        print("%-34s %12s %9d %9.1f us %5.1f%% %9.1f us %5.1f%% %9.1f us %5.1f%%" % (name, "%dx%d" % (rows, cols), bpe, same, byts / same * 1e-3 / 80,
              rot, byts / rot * 1e-3 / 80, eager, byts / eager * 1e-3 / 80), flush=True)
    def eager3(A, B, O):
        lib.binary(sma.OP_MUL, A, row, out=t1); lib.binary(sma.OP_ADD, t1, B, out=t2); lib.array_scalar(sma.OP_MUL, t2, half, out=O)
    report("(A * row + B) * 0.5", 12, lambda A, B, O: lib.chain_call(A, (sma.OP_MUL, row), (sma.OP_ADD, B), (sma.OP_MUL, half), out=O), eager3)
    def eager2(A, B, O):
        lib.binary(sma.OP_SUB, A, col, out=t1); lib.binary(sma.OP_DIV, t1, row, out=O)
    report("(A - col) / row", 8, lambda A, B, O: lib.chain_call(A, (sma.OP_SUB, col), (sma.OP_DIV, row), out=O), eager2)
    col2 = arr2(lib.uniform_f32(rows, 7, 0.5, 1.5), rows, 1)
    def eager_norm(A, B, O):
        lib.binary(sma.OP_SUB, A, col, out=t1); lib.binary(sma.OP_DIV, t1, col2, out=O)
    report("(A - mean_col) / std_col", 8, lambda A, B, O: lib.chain_call(A, (sma.OP_SUB, col), (sma.OP_DIV, col2), out=O), eager_norm)
    def eager_mul(A, B, O):
        lib.binary(sma.OP_SUB, A, col, out=t1); lib.binary(sma.OP_MUL, t1, row, out=O)
    report("(A - col) * row", 8, lambda A, B, O: lib.chain_call(A, (sma.OP_SUB, col), (sma.OP_MUL, row), out=O), eager_mul)
    def eager2b(A, B, O):
        lib.binary(sma.OP_ADD, A, B, out=t1); lib.array_scalar(sma.OP_MUL, t1, half, out=O)
    report("(A + B) * 0.5", 12, lambda A, B, O: lib.chain_call(A, (sma.OP_ADD, B), (sma.OP_MUL, half), out=O), eager2b)
    def eager1(A, B, O):
        lib.array_scalar(sma.OP_MUL, A, half, out=t1); lib.array_scalar(sma.OP_ADD, t1, half, out=O)
    report("A * 0.5 + 0.5", 8, lambda A, B, O: lib.chain_call(A, (sma.OP_MUL, half), (sma.OP_ADD, half), out=O), eager1)
    two = np.float32(2.0)
    def eager_sq(A, B, O):
        lib.binary(sma.OP_SUB, A, B, out=t1); lib.array_scalar(sma.OP_POW, t1, two, out=O)
    report("pow(A - B, 2)", 12, lambda A, B, O: lib.chain_call(A, (sma.OP_SUB, B), (sma.OP_POW, two), out=O), eager_sq)
    del sets, t1, t2, row, col
    lib.pool_trim()
