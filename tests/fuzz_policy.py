"""Soak for the size-dependent paths: random Ops on arrays of 4-70 MiB -- footprints on both sides of the stream-policy
thresholds (plain / nt loads, sc1 / nt stores), through the contiguous, scalar, row, column, tile, LDS, heavy-tile and fused
kernels -- against numpy, bit for bit (pow: within the 4-ULP bar).   usage: python tests/fuzz_policy.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplemath_amd as sma
from oracle import oracle as orc

DT = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    lib = sma.load()
    ops = {"add": (sma.OP_ADD, np.add), "sub": (sma.OP_SUB, np.subtract), "mul": (sma.OP_MUL, np.multiply)}
    for case in range(cases):
        dtn = rng.choice(["f32", "f32", "f64", "i32", "i64"])
        dt = DT[dtn]
        w = 16 // np.dtype(dt).itemsize
        mib = float(rng.choice([4, 9, 13.5, 14, 20, 21, 32, 43, 64, 70]))
        n_target = int(mib * (1 << 20) / np.dtype(dt).itemsize)
        cols = int(rng.choice([256, 1000, 1024, 4096, 3 * 224])) * (1 if rng.random() < 0.5 else w)
        rows = max(1, n_target // cols)
        n = rows * cols
        if dtn[0] == "f":
            a = rng.uniform(0.25, 4.0, n).astype(dt); b = rng.uniform(0.25, 4.0, n).astype(dt)
        else:
            info = np.iinfo(dt)
            a = rng.integers(info.min, info.max, n, dtype=dt); b = rng.integers(info.min, info.max, n, dtype=dt)
        da, db = lib.to_device(a), lib.to_device(b)
        kind = rng.choice(["contig", "scalar", "row", "col", "tile", "powrow", "powcol", "powscalar", "fused"])
        if rng.random() < 0.06:
            # a dense operand of >= 128 MiB against a small one that ignores the leading axes: the periodic route
            d1, d2, ch = int(rng.choice([7, 16, 28])), int(rng.choice([5, 16, 24])), int(rng.choice([2, 3, 4]))
            lead = (33 << 20) // (d1 * d2 * ch) + int(rng.integers(1, 9))
            xs = (lead, d1, d2, ch)
            xb = rng.uniform(0.25, 4.0, xs).astype(np.float32)
            ysm = rng.uniform(0.25, 4.0, (1, d1, 1, ch)).astype(np.float32)
            popn = str(rng.choice(["add", "sub", "mul"]))
            pop, pref = ops[popn]
            left = popn != "sub" and rng.random() < 0.5
            dxb, dys = lib.to_device(xb), lib.to_device(ysm)
            gotp = (lib.binary(pop, dys, dxb) if left else lib.binary(pop, dxb, dys)).numpy()
            wantp = pref(ysm, xb) if left else pref(xb, ysm)
            assert np.array_equal(gotp, wantp), f"case {case}: periodic {popn} {xs} left={left}"
            del dxb, dys, xb, gotp, wantp
            lib.pool_trim()
            continue
        opn = rng.choice(list(ops))
        op, ref = ops[opn]
        A = a.reshape(rows, cols)
        view = lambda d, shape, strides, off=0: sma.DeviceArray(lib, d.base_ptr, dt, shape, strides, off, d._owner)
        what = f"case {case}: {dtn} {kind} {opn} ({rows},{cols}) {mib} MiB"
        with np.errstate(all="ignore"):
            if kind == "contig":
                got, want = lib.contiguous(op, da, db).numpy(), ref(a, b)
            elif kind == "scalar":
                s = dt(3) if dtn[0] == "i" else dt(1.75)
                got, want = lib.array_scalar(op, da, s).numpy(), ref(a, s)
            elif kind == "row":
                r = b[:cols].reshape(1, cols)
                got, want = lib.binary(op, view(da, (rows, cols), (cols, 1)), view(db, (1, cols), (cols, 1))).numpy(), ref(A, r)
            elif kind == "col":
                c = b[:rows].reshape(rows, 1)
                got, want = lib.binary(op, view(da, (rows, cols), (cols, 1)), view(db, (rows, 1), (1, 1))).numpy(), ref(A, c)
            elif kind == "tile":
                got = lib.binary(op, view(da, (cols, rows), (1, cols)), view(db, (cols, rows), (rows, 1))).numpy()
                want = ref(A.T, b.reshape(cols, rows))
            elif kind == "fused":
                k = n // 2
                sub = lambda d: view(d, (k,), (1,))
                got = lib.fused(sma.OP_ADD, sma.OP_MUL, sub(da), sub(db), sub(da)).numpy()
                want = (a[:k] + b[:k]) * a[:k]
            else:  # pow forms: floats only
                if dtn[0] != "f":
                    continue
                if kind == "powscalar":
                    got = lib.array_scalar(sma.OP_POW, da, dt(2.5)).numpy()
                    want = np.power(a.astype(np.longdouble), np.longdouble(2.5)).astype(dt)
                elif kind == "powrow":
                    e = (b[:cols] - 2.0).reshape(1, cols)
                    got = lib.binary(sma.OP_POW, view(da, (rows, cols), (cols, 1)), lib.to_device(e)).numpy()
                    want = np.power(A.astype(np.longdouble), e.astype(np.longdouble)).astype(dt)
                else:
                    e = (b[:rows] - 2.0).reshape(rows, 1)
                    got = lib.binary(sma.OP_POW, view(da, (rows, cols), (cols, 1)), lib.to_device(e)).numpy()
                    want = np.power(A.astype(np.longdouble), e.astype(np.longdouble)).astype(dt)
                if dt == np.float32:
                    assert orc.ulp_diff_f32(got.reshape(-1), want.reshape(-1)).max() <= 4, what
                else:
                    assert np.abs(got.reshape(-1).view(np.int64) - want.reshape(-1).view(np.int64)).max() <= 1, what
                continue
        assert np.array_equal(got.reshape(-1), np.asarray(want).reshape(-1)), what
        del da, db
        if case % 20 == 19:
            lib.pool_trim()
            print(f"{case + 1} cases", flush=True)
    print(f"ok: {cases} policy cases, seed {seed}")


if __name__ == "__main__":
    main()
