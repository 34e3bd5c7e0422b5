"""Randomised views against the oracle: sliced (with steps), permuted, broadcast operands of rank 1-5 through
smhip_elementwise, and random strided assignments through smhip_copy_strided (numpy is the specification there).

usage: python tests/fuzz_views.py [cases] [seed]      -- prints the first mismatch and exits 1, else "ok".
"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
from oracle import oracle as orc
from tests import util
from tests.golden import gen

DT = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64}


def random_view(rng, base):
    """A random basic-indexing view of `base` (slices with steps, optional axis permutation)."""
    sel = []
    for d in base.shape:
        step = int(rng.choice([1, 1, 1, 1, 2, 3]))
        lo = int(rng.integers(0, max(1, d // 3 + 1)))
        hi = int(rng.integers(min(d, lo + 1), d + 1))
        sel.append(slice(lo, hi, step))
    v = base[tuple(sel)]
    if v.ndim >= 2 and rng.random() < 0.4:
        v = np.transpose(v, rng.permutation(v.ndim))
    return v


def random_base(rng, dt, seed, kind):
    nd = int(rng.integers(1, 6))
    big_axis = int(rng.integers(0, nd))
    shape = [int(rng.integers(1, 10)) for _ in range(nd)]
    shape[big_axis] = int(rng.integers(1, 140))
    if nd >= 2 and rng.random() < 0.5:
        shape[int(rng.integers(0, nd))] = int(rng.integers(8, 80))
    return gen.gen(dt, int(np.prod(shape)), seed, kind).reshape(shape)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    lib = sma.load()
    oracle = orc.Oracle()
    ops = ["add", "sub", "mul", "div"]
    user_mul = lib.register_op("a * b")
    done = 0
    for t in range(cases):
        dtn = ["f32", "f64", "i32", "i64"][t % 4]
        dt = DT[dtn]
        op = ops[(t // 4) % 4]
        is_pow = dtn[0] == "f" and t % 20 >= 18  # some float cases: pow through the broadcast kernels (positive bases, <= 4 ULP)
        if is_pow:
            op = "pow"
        a = random_base(rng, dt, 10000 + t, "positive" if is_pow else "uniform")
        av = random_view(rng, a)
        # b: same shape as av's result (a view of its own base) or a broadcastable reduction of it
        bshape = [d if rng.random() < 0.65 else 1 for d in av.shape][int(rng.integers(0, av.ndim)):] or [1]
        pad = [int(rng.integers(0, 4)) for _ in bshape]
        bb = gen.gen(dt, int(np.prod([d + p for d, p in zip(bshape, pad)])), 20000 + t, "nonzero" if (dtn[0] == "i" and op == "div") else "uniform")
        bb = bb.reshape([d + p for d, p in zip(bshape, pad)])
        bv = bb[tuple(slice(int(rng.integers(0, p + 1)), None) for p in pad)]
        bv = bv[tuple(slice(0, d) for d in bshape)]
        if rng.random() < 0.3 and not is_pow:
            av, bv, a, bb = bv, av, bb, a  # small operand on the left
            if dtn[0] == "i" and op == "div":
                op = "mul"
        da, db = lib.to_device(a), lib.to_device(bb)
        opid = user_mul if (op == "mul" and t % 3 == 0) else sma.OPS[op]  # a third of the multiplies: the hipRTC path, same kernels
        try:
            got = lib.binary(opid, da.view_like(av, a), db.view_like(bv, bb)).numpy()
        except sma.SmhipError as e:
            print(f"ERROR case {t} seed {seed}: {dtn} {op} a{av.shape} strides {tuple(s // av.itemsize for s in av.strides)} "
                  f"b{bv.shape} strides {tuple(s // bv.itemsize for s in bv.strides)}: {e}")
            sys.exit(1)
        want = oracle.binary(orc.OPS[op], av, bv)
        try:
            if is_pow:
                ulps = (orc.ulp_diff_f32 if dtn == "f32" else orc.ulp_diff_f64)(got, want)
                assert ulps.max() <= 4, f"pow off by {ulps.max()} ULP"
            else:
                util.assert_same_bits(got, want, "")
        except AssertionError as e:
            print(f"MISMATCH case {t} seed {seed}: {dtn} {op} a{av.shape} strides {tuple(s // av.itemsize for s in av.strides)} "
                  f"b{bv.shape} strides {tuple(s // bv.itemsize for s in bv.strides)}\n{e}")
            sys.exit(1)
        # the dense copy of the view itself (SMHIP_OP_LEFT: contiguous(), with its repeat / deinterleave / tile special cases)
        a_lo = a.__array_interface__["data"][0]
        if a_lo <= av.__array_interface__["data"][0] < a_lo + a.nbytes:  # av is a view of a (not the swapped small operand)
            dense = lib.binary(sma.OP_LEFT, da.view_like(av, a), lib.to_device(np.zeros(1, dtype=dt))).numpy()
            if not np.array_equal(dense.view(np.uint8), np.ascontiguousarray(av).view(np.uint8)):
                print(f"LEFT MISMATCH case {t} seed {seed}: {dtn} a{av.shape} strides {tuple(s // av.itemsize for s in av.strides)}")
                sys.exit(1)
        # assignment: dst view of a  <-  source view broadcast to it
        dst_base = random_base(rng, dt, 30000 + t, "uniform")
        dv = random_view(rng, dst_base)
        src = gen.gen(dt, int(np.prod(dv.shape)) * 2 + 7, 40000 + t, "uniform")
        sshape = [d if rng.random() < 0.8 else 1 for d in dv.shape]
        sv = src[3:3 + int(np.prod(sshape))].reshape(sshape)
        want = dst_base.copy()
        wv = want[tuple(slice(None) for _ in want.shape)]
        # rebuild the same view on `want`
        off = (dv.__array_interface__["data"][0] - dst_base.__array_interface__["data"][0])
        wview = np.lib.stride_tricks.as_strided(want.reshape(-1)[off // want.itemsize:], shape=dv.shape, strides=dv.strides)
        # overlapping destinations (a permuted view never overlaps itself; steps keep elements distinct)
        wview[...] = sv
        ddst, dsrc = lib.to_device(dst_base), lib.to_device(src)
        lib.assign(ddst.view_like(dv, dst_base), dsrc.view_like(sv, src))
        if not np.array_equal(ddst.numpy().view(np.uint8), want.view(np.uint8)):
            print(f"ASSIGN MISMATCH case {t} seed {seed}: {dtn} dst{dv.shape} strides {tuple(s // dv.itemsize for s in dv.strides)} src{sv.shape}")
            sys.exit(1)
        done += 1
    print(f"ok: {done} elementwise + {done} assignment cases, seed {seed}")


if __name__ == "__main__":
    main()
