"""Randomised 1-D forms against the oracle: contiguous, array-scalar, fused and reduction entry points on random lengths
(every tail) and random element offsets (every misalignment), all element types.   usage: fuzz_flat.py [cases] [seed]"""
import sys
sys.path.insert(0, "/root/repo")
import math
import numpy as np
import simplemath_amd as sma
from oracle import oracle as orc
from tests import util
from tests.golden import gen

DT = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    lib = sma.load()
    oracle = orc.Oracle()
    ops = ["add", "sub", "mul", "div"]
    for t in range(cases):
        dtn = ["f32", "f64", "i32", "i64"][t % 4]
        dt = DT[dtn]
        isint = dtn[0] == "i"
        n = int(rng.choice([rng.integers(1, 20), rng.integers(1, 5000), rng.integers(1, 300000), 1024 * int(rng.integers(1, 300)) + int(rng.integers(0, 4))]))
        oa, ob = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        A = gen.gen(dt, n + 8, 50000 + t, "uniform"); B = gen.gen(dt, n + 8, 60000 + t, "nonzero" if isint else "uniform")
        Cc = gen.gen(dt, n + 8, 70000 + t, "nonzero" if isint else "uniform")
        a, b, c = A[oa:oa + n], B[ob:ob + n], Cc[1:1 + n]
        dA, dB, dC = lib.to_device(A), lib.to_device(B), lib.to_device(Cc)
        da, db, dc = dA.view_like(a, A), dB.view_like(b, B), dC.view_like(c, Cc)
        op = ops[(t // 4) % 4]
        what = f"case {t} seed {seed} {dtn} {op} n={n} offsets {oa},{ob}"
        try:
            util.assert_same_bits(lib.contiguous(sma.OPS[op], da, db).numpy(), oracle.contiguous(orc.OPS[op], np.ascontiguousarray(a), np.ascontiguousarray(b)), what)
            sc = dt(3) if isint else dt(1.75)
            util.assert_same_bits(lib.array_scalar(sma.OPS[op], da, sc).numpy(), oracle.array_scalar(orc.OPS[op], np.ascontiguousarray(a), sc), what + " scalar")
            op2 = ops[(t // 16) % 3]  # add/sub/mul as the second stage
            two = oracle.contiguous(orc.OPS[op2], oracle.contiguous(orc.OPS[op], np.ascontiguousarray(a), np.ascontiguousarray(b)), np.ascontiguousarray(c))
            util.assert_same_bits(lib.fused(sma.OPS[op], sma.OPS[op2], da, db, dc).numpy(), two, what + f" fused {op2}")
            # reductions
            s_gpu, s_ref = lib.sum(da), oracle.sum(np.ascontiguousarray(a))
            d_gpu = lib.dot(da, db)
            if isint:
                assert s_gpu == s_ref, (what, s_gpu, s_ref)
                want = oracle.dot(np.ascontiguousarray(a), np.ascontiguousarray(b))
                assert int(d_gpu) == int(want), (what, d_gpu, want)
            else:
                scale = float(np.abs(a.astype(np.float64)).sum()) + 1.0
                assert abs(s_gpu - s_ref) <= n * 2.0 ** -52 * scale, (what, s_gpu, s_ref)
                exact = math.fsum((a.astype(np.float64) * b.astype(np.float64)).tolist())
                dscale = float(np.abs(a.astype(np.float64) * b.astype(np.float64)).sum()) + 1.0
                tol = n * 2.0 ** -52 * dscale + (2.0 ** -23 if dtn == "f32" else 2.0 ** -52) * abs(exact)
                assert abs(float(d_gpu) - exact) <= tol, (what, float(d_gpu), exact, tol)
        except (AssertionError, sma.SmhipError) as e:
            print("MISMATCH", what, e)
            sys.exit(1)
    print(f"ok: {cases} flat cases, seed {seed}")


if __name__ == "__main__":
    main()
