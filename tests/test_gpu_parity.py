"""GPU parity: the HIP path, called through the C ABI (include/smhip.h), against
(1) the reference's own gtest cases, (2) the fixtures recorded from the compiled
reference, (3) the oracle on seeded random inputs, (4) size-independent
properties at BASELINE.json's full sizes, (5) edge cases.

Bars (BASELINE north_star): int32/int64 bit-exact; f32/f64 + - * / bit-exact
(one correctly rounded IEEE op on both sides; a NaN result's payload is not
compared); f32 pow within 4 ULP of the correctly rounded value (budget 2, measured 1);
reductions against the fp64 oracle with the tolerance stated at each test.
"""
import ctypes as C

import numpy as np
import pytest

import simplemath_amd as sma
from oracle import oracle as orc
from tests import util
from tests.golden import cases, gen

pytestmark = pytest.mark.gpu
DT = cases.DT
POW_ULP = 4  # BASELINE north_star: "within ... 4 ULP (pow/exp) for float"


# ------------------------------------------------------ 1. the reference's KATs
@pytest.mark.parametrize("k", util.load_kat(), ids=lambda k: k["name"])
def test_reference_kat(smhip, k):
    dt = DT[k["dtype"]]
    abase, av = util.kat_array(k["a"], dt)
    _, exp = util.kat_array(k["expected"], dt)
    da = smhip.to_device(abase)
    if "scalar" in k:
        got = smhip.array_scalar(sma.OPS[k["op"]], da, k["scalar"]).numpy()
    else:
        bbase, bv = util.kat_array(k["b"], dt)
        db = smhip.to_device(bbase)
        got = smhip.binary(sma.OPS[k["op"]], da.view_like(av, abase), db.view_like(bv, bbase)).numpy()
    assert list(got.shape) == list(exp.shape)
    if k["cmp"] == "eq":
        assert np.array_equal(got, exp)
    elif k["cmp"] == "float_eq":
        assert orc.ulp_diff_f32(got, exp).max() <= 4  # EXPECT_FLOAT_EQ
    else:
        assert orc.ulp_diff_f64(got, exp).max() <= 4  # EXPECT_DOUBLE_EQ


# -------------------------------------- 2. fixtures recorded from the reference
def _check_fixture(store, cid, out, big=False):
    flat = np.ascontiguousarray(out).reshape(-1)
    exp = store[f"{cid}/out"]
    if big:
        util.assert_same_bits(np.concatenate([flat[:64], flat[-64:]]), exp, cid)
        if flat.dtype.kind != "f" or not np.isnan(flat).any():
            assert cases.digest(flat) == bytes(store[f"{cid}/sha"]).hex(), cid
    else:
        util.assert_same_bits(flat, exp, cid)


def test_golden_contiguous(smhip):
    st = util.load_npz("contiguous.npz")
    for c in cases.contiguous_cases():
        a, b = cases.contiguous_inputs(c)
        out = smhip.contiguous(sma.OPS[c["op"]], smhip.to_device(a), smhip.to_device(b)).numpy()
        _check_fixture(st, c["id"], out)
        # the same through the general entry point (calculate.h:10-11's fast-path dispatch)
        da, db = smhip.to_device(a), smhip.to_device(b)
        out2 = smhip.binary(sma.OPS[c["op"]], da, db).numpy()
        _check_fixture(st, c["id"], out2)


@pytest.mark.parametrize("c", cases.broadcast_cases(), ids=lambda c: c["id"])
def test_golden_broadcast(smhip, oracle, c):
    st = util.load_npz("broadcast.npz")
    abase, av, bbase, bv = cases.broadcast_inputs(c)
    da, db = smhip.to_device(abase), smhip.to_device(bbase)
    out = smhip.binary(sma.OPS[c["op"]], da.view_like(av, abase), db.view_like(bv, bbase))
    assert list(out.shape) == list(st[f"{c['id']}/shape"])
    got = out.numpy()
    _check_fixture(st, c["id"], got, big=c["big"])
    util.assert_same_bits(got, oracle.binary(orc.OPS[c["op"]], av, bv), c["id"] + " vs oracle")


def test_golden_scalar(smhip):
    st = util.load_npz("scalar.npz")
    for c in cases.scalar_cases():
        a = cases.scalar_input(c)
        out = smhip.array_scalar(sma.OPS[c["op"]], smhip.to_device(a), c["scalar"]).numpy()
        _check_fixture(st, c["id"], out)


def test_golden_bench_sizes(smhip):
    """Reference-recorded fixtures at the sizes the reference benchmarks and tests at: contiguous add /
    div / mul at N = 1 000 000 (benchmark/add.cpp:21-29: ones, and seeded), the scalar forms at 100 003 and 1 000 000,
    int pow on 2 000 000 elements (tests/pow.cpp:46-61).  Head, tail and the sha256 of the whole output."""
    st = util.load_npz("bench_sizes.npz")
    for c in cases.bench_size_cases():
        a, b = cases.bench_size_inputs(c)
        da = smhip.to_device(a)
        if c["kind"] == "contig":
            db = smhip.to_device(b)
            _check_fixture(st, c["id"], smhip.contiguous(sma.OPS[c["op"]], da, db).numpy(), big=True)
            _check_fixture(st, c["id"], smhip.binary(sma.OPS[c["op"]], da, db).numpy(), big=True)  # through element_wise_op's dispatch
        elif c["kind"] == "scalar":
            _check_fixture(st, c["id"], smhip.array_scalar(sma.OPS[c["op"]], da, c["scalar"]).numpy(), big=True)
        else:
            # n is a multiple of 8: no libm tail in the reference (calculate.h:166-168), the vector body everywhere
            assert c["n"] % 8 == 0
            _check_fixture(st, c["id"], smhip.array_scalar(sma.OP_POW, da, c["exp"]).numpy(), big=True)


def _pow_fits_i32(base, e):
    if e < 0 and base == 0:
        return False  # the reference's libm tail turns 0^negative = inf into INT_MIN; its vector body gives 0
    if e < 0 or abs(base) <= 1:
        return True
    if e > 31:
        return False
    return abs(base ** e) < 2 ** 31


def test_golden_ipow(smhip, oracle):
    st = util.load_npz("ipow.npz")
    for c in cases.ipow_cases():
        a = cases.ipow_input(c)
        got = smhip.array_scalar(sma.OP_POW, smhip.to_device(a), c["exp"]).numpy()
        ref = st[f"{c['id']}/out"]
        nb = c["n"] - c["n"] % 8
        # vector body of the reference (crafted_pow.h:54-103): bit-exact
        assert np.array_equal(got[:nb], ref[:nb]), c["id"]
        # tail: the reference switches to std::pow -> double -> int (calculate.h:166-168), which
        # differs from its own body only where the true power overflows int32;
        # the HIP path uses the body's definition everywhere
        body = oracle.array_scalar(orc.POW, a, c["exp"], int_pow_tail_libm=False)
        assert np.array_equal(got, body), c["id"]
        fits = np.array([_pow_fits_i32(int(x), c["exp"]) for x in a[nb:]], dtype=bool)
        assert np.array_equal(got[nb:][fits], ref[nb:][fits]), c["id"]


def test_golden_dot(smhip, oracle):
    st = util.load_npz("dot.npz")
    for c in cases.dot_cases():
        a, b = cases.dot_inputs(c)
        got = smhip.dot(smhip.to_device(a), smhip.to_device(b))
        ref = st[f"{c['id']}/out"][0]
        if a.dtype.kind == "i":
            assert got == ref, c["id"]  # wrapping arithmetic: order-independent, bit-exact
        else:
            exact = float(oracle.dot(a, b, lane_order=False))
            scale = float(np.abs(a.astype(np.float64) * b.astype(np.float64)).sum())
            eps = np.finfo(a.dtype).eps
            # fp64 fma chain: one final rounding to the result type + n roundings at 2^-53 of the running sum
            assert abs(float(got) - exact) <= eps * abs(exact) + c["n"] * 2.0 ** -53 * scale, c["id"]
            if a.dtype == np.float32:
                # and at least as close to the truth as the reference's f32 lane accumulators are
                assert abs(float(got) - exact) <= abs(float(ref) - exact) + eps * abs(exact), c["id"]


def test_golden_powf(smhip):
    st = util.load_npz("powf.npz")
    worst = 0
    for c in cases.powf_cases():
        a = cases.powf_input(c)
        got = smhip.array_scalar(sma.OP_POW, smhip.to_device(a), np.float32(c["exp"])).numpy()
        d_exact = orc.ulp_diff_f32(got, st[f"{c['id']}/exact"])
        d_libm = orc.ulp_diff_f32(got, st[f"{c['id']}/out"])  # PowOp<float>::apply = glibc powf
        assert d_exact.max() <= POW_ULP and d_libm.max() <= POW_ULP, (c["id"], int(d_exact.max()), int(d_libm.max()))
        # specials must agree in kind and sign, not just in distance
        ref = st[f"{c['id']}/out"]
        sp = ~np.isfinite(ref) | (ref == 0)
        assert np.array_equal(np.isnan(got[sp]), np.isnan(ref[sp])), c["id"]
        nn = sp & ~np.isnan(ref)
        util.assert_same_bits(got[nn], ref[nn], c["id"] + " specials")
        worst = max(worst, int(d_exact.max()))
    assert worst <= 2  # sm_pow.h's own error budget (fp64 log2 chain + f32 exp stage); 1 observed


# ---------------------------------------------- 3. oracle on seeded random inputs
@pytest.mark.parametrize("dt", ["f32", "f64", "i32", "i64"])
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "pow"])
def test_contiguous_vs_oracle(smhip, oracle, dt, op):
    for n in (1, 3, 4, 5, 257, 4096, 100003, 1 << 20):
        a = gen.gen(DT[dt], n, 100 + n, "mixed")
        kind_b = "nonzero" if (op == "div" and dt[0] == "i") else "mixed"
        b = gen.gen(DT[dt], n, 200 + n, kind_b)
        if op == "pow":
            if dt[0] == "i":
                b = (b % 7).astype(DT[dt]) - DT[dt](2)
            else:
                a, b = gen.gen(DT[dt], n, 100 + n, "positive"), gen.gen(DT[dt], n, 200 + n, "uniform")
        got = smhip.contiguous(sma.OPS[op], smhip.to_device(a), smhip.to_device(b)).numpy()
        want = oracle.contiguous(orc.OPS[op], a, b)
        if op == "pow" and dt == "f64":
            assert orc.ulp_diff_f64(got, want).max() <= 1  # vs glibc pow, itself < 1 ULP
        elif op == "pow" and dt == "f32":
            with np.errstate(all="ignore"):
                exact = np.power(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
            assert orc.ulp_diff_f32(got, exact).max() <= POW_ULP
            assert orc.ulp_diff_f32(got, want).max() <= POW_ULP
        else:
            util.assert_same_bits(got, want, f"{dt} {op} n={n}")


def test_f64_pow(smhip, oracle):
    """PowOp<double> (sm_pow64.h): within 1 ULP of glibc pow (the reference's arithmetic) incl. specials."""
    for kind, n in (("positive", 100003), ("mixed", 4096), ("wide", 65536)):
        a = gen.gen(np.float64, n, 1, kind)
        for e in (2.5, -1.5, 0.5, 3.0, 700.0, -0.001, 0.0, float("inf"), float("nan")):
            got = smhip.array_scalar(sma.OP_POW, smhip.to_device(a), e).numpy()
            want = oracle.array_scalar(orc.POW, a, e)
            assert orc.ulp_diff_f64(got, want).max() <= 1, (kind, e)
    b = gen.gen(np.float64, 4096, 2, "uniform") * 10
    a = gen.gen(np.float64, 4096, 3, "positive")
    got = smhip.contiguous(sma.OP_POW, smhip.to_device(a), smhip.to_device(b)).numpy()
    assert orc.ulp_diff_f64(got, oracle.contiguous(orc.POW, a, b)).max() <= 1


def test_pow_exact_exponents(smhip, oracle):
    """sm::pow(a, s) for s in {2, 1, -1, 0.5}: one correctly rounded IEEE operation each (x*x, x, 1/x, sqrt x with pow's
    answers at -0 and -inf), so the result must equal the correctly rounded power bit for bit -- specials, denormals,
    overflow and all -- and agree with libm's pow (the reference's PowOp<T>::apply) to <= 1 ULP."""
    for dt, ulp in ((np.float32, orc.ulp_diff_f32), (np.float64, orc.ulp_diff_f64)):
        a = np.concatenate([gen.gen(dt, 5000, 101, "mixed"), gen.gen(dt, 3001, 102, "positive"),
                            np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 4.0, 1e-40, -1e-40, 3e38, -3e38], dtype=dt)])
        da = smhip.to_device(a)
        with np.errstate(all="ignore"):
            sq = np.sqrt(a)
            sq[a == 0] = 0.0          # pow(-0, 0.5) = +0
            sq[np.isneginf(a)] = np.inf  # pow(-inf, 0.5) = +inf
            want = {2.0: a * a, 1.0: a.copy(), -1.0: (dt(1) / a), 0.5: sq}
        for y, w in want.items():
            got = smhip.array_scalar(sma.OP_POW, da, dt(y)).numpy()
            util.assert_same_bits(got, w, f"{np.dtype(dt)} ^ {y}")
            lib = oracle.array_scalar(orc.POW, a, dt(y))
            ok = ~(np.isnan(got) | np.isnan(lib))
            assert np.array_equal(np.isnan(got), np.isnan(lib)), (np.dtype(dt), y)
            assert ulp(got[ok], lib[ok]).max() <= 1, (np.dtype(dt), y)


def test_pow_f64_half_integer_exponents(smhip, oracle):
    """sm::pow(a, s) on doubles with s a multiple of one half, |s| <= 8: the double-double product chain of sm_pow64.h
    (pow_halfint) instead of exp(s log a).  Every such exponent against libm's pow (the reference's PowOp<double>::apply,
    pow.h:8-10): <= 1 ULP, NaNs in the same places, zeros and infinities with the same sign -- over mixed-sign values,
    subnormals, values whose power overflows or underflows, and the special values; odd tails included."""
    a = np.concatenate([gen.gen(np.float64, 6001, 111, "mixed"), gen.gen(np.float64, 4000, 112, "positive"),
                        np.exp2(np.linspace(-1074, 1023, 4003)), -np.exp2(np.linspace(-300, 300, 1001)),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 4.0, 5e-324, -5e-324, 1e-310, 1.7976931348623157e308,
                                  -1.7976931348623157e308, 2.2250738585072014e-308], dtype=np.float64)])
    da = smhip.to_device(a)
    for m2 in range(-16, 17):
        if m2 == 0:
            continue
        y = np.float64(m2 * 0.5)
        got = smhip.array_scalar(sma.OP_POW, da, y).numpy()
        with np.errstate(all="ignore"):
            want = oracle.array_scalar(orc.POW, a, y)
        assert np.array_equal(np.isnan(got), np.isnan(want)), y
        ok = ~np.isnan(want)
        same_kind = ((got[ok] == 0) == (want[ok] == 0)) & (np.isinf(got[ok]) == np.isinf(want[ok]))
        # a power that libm rounds to 0 / the smallest subnormal (or to inf / the largest double) may differ by that one ULP
        assert orc.ulp_diff_f64(got[ok], want[ok]).max() <= 1, (y, a[ok][np.argmax(orc.ulp_diff_f64(got[ok], want[ok]))])
        assert np.array_equal(np.signbit(got[ok]), np.signbit(want[ok])), y
        assert same_kind.mean() > 0.999, y


def test_pow_f64_one_exponent(smhip, oracle):
    """sm::pow(a, s) on doubles with one exponent up to 1024 in magnitude that is not a half-integer: sm_pow64.h's pow_core_u
    (level 2 up to 16, level 1 up to 1024; beyond that the general form).  Against libm's pow (PowOp<double>::apply, pow.h:8-10):
    <= 1 ULP, NaNs in the same places, same signs -- mixed-sign values, subnormals, powers that overflow / underflow, specials,
    odd tails and a size that spans several workgroups."""
    a = np.concatenate([gen.gen(np.float64, 6001, 121, "mixed"), gen.gen(np.float64, 70001, 122, "positive"),
                        np.exp2(np.linspace(-1074, 1023, 4003)), -np.exp2(np.linspace(-300, 300, 1001)),
                        1.0 + np.linspace(-0.05, 0.05, 3001),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 4.0, 5e-324, -5e-324, 1e-310, 1.7976931348623157e308,
                                  -1.7976931348623157e308, 2.2250738585072014e-308], dtype=np.float64)])
    da = smhip.to_device(a)
    for y in (2.7, 1.0 / 3, 0.3333, 8.5, -13.37, 15.99, 16.0, -16.0, 1e-3, 1e-300, 16.000000000000004, 37.75, 100.25, -700.3, 1024.0, -1024.0,
              1024.0000000000002, 5000.1, 17.0, 33.0):
        got = smhip.array_scalar(sma.OP_POW, da, np.float64(y)).numpy()
        with np.errstate(all="ignore"):
            want = oracle.array_scalar(orc.POW, a, np.float64(y))
        assert np.array_equal(np.isnan(got), np.isnan(want)), y
        ok = ~np.isnan(want)
        d = orc.ulp_diff_f64(got[ok], want[ok])
        assert d.max() <= 1, (y, a[ok][np.argmax(d)])
        assert np.array_equal(np.signbit(got[ok]), np.signbit(want[ok])), y


@pytest.mark.parametrize("dt", ["f32", "f64", "i32", "i64"])
def test_array_scalar_vs_oracle(smhip, oracle, dt):
    for op in ("add", "sub", "mul", "div"):
        for n in (1, 7, 1000, 65537):
            a = gen.gen(DT[dt], n, 300 + n, "mixed")
            s = {"f": 1.5, "i": 3}[np.dtype(DT[dt]).kind]
            got = smhip.array_scalar(sma.OPS[op], smhip.to_device(a), s).numpy()
            util.assert_same_bits(got, oracle.array_scalar(orc.OPS[op], a, s), f"{dt} {op} {n}")


def test_random_broadcast_vs_oracle(smhip, oracle):
    rng = np.random.default_rng(2024)
    for trial in range(120):
        nd = int(rng.integers(1, 7))
        shape = [int(rng.integers(1, 9)) for _ in range(nd)]
        if trial % 5 == 0:
            shape[-1] = int(rng.integers(16, 200))  # reach the row kernel
        ash = [d if rng.random() < 0.7 else 1 for d in shape]
        bsh = [d if rng.random() < 0.7 else 1 for d in shape][int(rng.integers(0, nd)):]
        dtn = ["f32", "f64", "i32", "i64"][trial % 4]
        op = cases.BINOPS[(trial // 4) % 4]
        a = gen.gen(DT[dtn], int(np.prod(ash)), 7000 + trial, "mixed").reshape(ash)
        kb = "nonzero" if (dtn[0] == "i" and op == "div") else "mixed"
        b = gen.gen(DT[dtn], max(int(np.prod(bsh)), 1), 8000 + trial, kb).reshape(bsh if bsh else (1,))
        av, bv = a, b
        if trial % 7 == 3 and a.ndim >= 2:
            av = a.transpose()  # SMArray::transpose()
            b = gen.gen(DT[dtn], int(np.prod(av.shape)), 8000 + trial, kb).reshape(av.shape)
            bv = b
        da, db = smhip.to_device(a), smhip.to_device(b)
        got = smhip.binary(sma.OPS[op], da.view_like(av, a), db.view_like(bv, b)).numpy()
        want = oracle.binary(orc.OPS[op], av, bv)
        util.assert_same_bits(got, want, f"trial {trial} {dtn} {op} {av.shape} {bv.shape}")


def test_broadcast_pow_vs_oracle(smhip, oracle):
    """pow through the row / gather / device-scalar kernels (elementwise exponents, not just sm::pow's scalar)."""
    base = gen.gen(np.float32, 37 * 64, 21, "positive").reshape(37, 64)
    for eshape in ((1, 64), (37, 1), (37, 64), (1,), (64,)):
        e = gen.gen(np.float32, int(np.prod(eshape)), 22, "uniform").reshape(eshape)
        got = smhip.binary(sma.OP_POW, smhip.to_device(base), smhip.to_device(e)).numpy()
        with np.errstate(all="ignore"):
            exact = np.power(base.astype(np.float64), e.astype(np.float64)).astype(np.float32)
        assert orc.ulp_diff_f32(got, exact).max() <= POW_ULP, eshape
        assert orc.ulp_diff_f32(got, oracle.binary(orc.POW, base, e)).max() <= POW_ULP, eshape
    bt = base.T  # transposed base: gather kernel
    db = smhip.to_device(base)
    e = gen.gen(np.float32, 37, 23, "uniform").reshape(1, 37)
    got = smhip.binary(sma.OP_POW, db.view_like(bt, base), smhip.to_device(e)).numpy()
    with np.errstate(all="ignore"):
        exact = np.power(bt.astype(np.float64), e.astype(np.float64)).astype(np.float32)
    assert orc.ulp_diff_f32(got, exact).max() <= POW_ULP
    ib = (gen.gen(np.int32, 37 * 64, 24, "uniform") % 20).astype(np.int32).reshape(37, 64)
    ie = (gen.gen(np.int32, 64, 25, "uniform") % 9 - 2).astype(np.int32).reshape(1, 64)
    got = smhip.binary(sma.OP_POW, smhip.to_device(ib), smhip.to_device(ie)).numpy()
    assert np.array_equal(got, oracle.binary(orc.POW, ib, ie))


def test_pow_by_a_row_or_column_of_exponents(smhip):
    """Config 3's shape with pow -- a dense base against ONE ROW or ONE COLUMN of exponents -- takes the heavy tile kernel
    (KIND 3 / 4 of flat_tile_kernel): f32 and f64, extents that end inside a tile, that span many tiles, a single row, and
    special exponents in the broadcast operand; against numpy's correctly rounded power."""
    rng = np.random.default_rng(91)
    for dt, ulp_of, bar in ((np.float32, orc.ulp_diff_f32, POW_ULP), (np.float64, None, 1)):
        w = 16 // np.dtype(dt).itemsize
        for rows, cols in ((37, 16 * w), (300, 256 * w), (1, 64 * w), (1025, 4 * w), (513, 130 * w)):
            base = rng.uniform(0.05, 30.0, (rows, cols)).astype(dt)
            for eshape in ((1, cols), (rows, 1)):
                e = rng.uniform(-3.0, 3.0, eshape).astype(dt)
                flat = e.reshape(-1)
                for k, special in enumerate((0.0, 1.0, 2.0, -1.0, 0.5, 3.0, -0.0)):
                    flat[(k * 5) % flat.size] = special
                got = smhip.binary(sma.OP_POW, smhip.to_device(base), smhip.to_device(e)).numpy()
                with np.errstate(all="ignore"):
                    exact = np.power(base.astype(np.longdouble), e.astype(np.longdouble)).astype(dt)
                if ulp_of is not None:
                    assert ulp_of(got, exact).max() <= bar, (dt, rows, cols, eshape)
                else:
                    d = np.abs(got.view(np.int64) - exact.view(np.int64))
                    assert d.max() <= bar, (dt, rows, cols, eshape)


def test_transposed_and_permuted_views(smhip, oracle):
    """Operands whose contiguous axis is not the output's inner axis go through the LDS tile
    kernel; patch edges, both operands transposed, permuted 3-D / 4-D views,
    broadcasting on top, every dtype."""
    for dtn, op in (("f32", "add"), ("f64", "mul"), ("i32", "sub"), ("i64", "add"), ("f32", "div"), ("f32", "pow")):
        dt = DT[dtn]
        kind = "positive" if op == "pow" else "uniform"
        for (r, c) in ((100, 70), (64, 64), (65, 129), (16, 1000), (257, 33)):
            a = gen.gen(dt, r * c, 31, kind).reshape(r, c)
            b = gen.gen(dt, r * c, 32, "nonzero" if (dtn[0] == "i") else "uniform").reshape(c, r)
            da, db = smhip.to_device(a), smhip.to_device(b)
            for av, bv in ((a.T, b), (a, b.T), (a.T, b.T.T.T)):
                if av.shape != bv.shape:
                    continue
                got = smhip.binary(sma.OPS[op], da.view_like(av, a), db.view_like(bv, b)).numpy()
                want = oracle.binary(orc.OPS[op], av, bv)
                if op == "pow":
                    assert orc.ulp_diff_f32(got, want).max() <= POW_ULP
                else:
                    util.assert_same_bits(got, want, f"{dtn} {op} {av.shape}")
        # both transposed (a.T op c.T), same storage order
        a = gen.gen(dt, 90 * 75, 33, kind).reshape(90, 75)
        c = gen.gen(dt, 90 * 75, 34, "nonzero" if dtn[0] == "i" else "uniform").reshape(90, 75)
        got = smhip.binary(sma.OPS[op], smhip.to_device(a).view_like(a.T, a), smhip.to_device(c).view_like(c.T, c)).numpy()
        want = oracle.binary(orc.OPS[op], a.T, c.T)
        if op == "pow":
            assert orc.ulp_diff_f32(got, want).max() <= POW_ULP
        else:
            util.assert_same_bits(got, want, f"{dtn} {op} both-T")
    # permuted 3-D / 4-D views (beyond transpose()'s full reversal) with broadcasting on top
    a = gen.gen(np.float32, 20 * 48 * 33, 35, "uniform").reshape(20, 48, 33)
    da = smhip.to_device(a)
    for perm in ((1, 0, 2), (2, 1, 0), (0, 2, 1), (2, 0, 1), (1, 2, 0)):
        av = np.transpose(a, perm)
        bshape = [1 if k == 1 else d for k, d in enumerate(av.shape)]
        b = gen.gen(np.float32, int(np.prod(bshape)), 36, "uniform").reshape(bshape)
        got = smhip.binary(sma.OP_MUL, da.view_like(av, a), smhip.to_device(b)).numpy()
        util.assert_same_bits(got, oracle.binary(orc.MUL, av, b), f"perm {perm}")
    a4 = gen.gen(np.int32, 3 * 40 * 5 * 36, 37, "uniform").reshape(3, 40, 5, 36)
    av = np.transpose(a4, (0, 3, 2, 1))
    b4 = gen.gen(np.int32, 36 * 40, 38, "uniform").reshape(36, 1, 40)
    got = smhip.binary(sma.OP_ADD, smhip.to_device(a4).view_like(av, a4), smhip.to_device(b4)).numpy()
    assert np.array_equal(got, oracle.binary(orc.ADD, av, b4))


def test_dense_against_a_handful_of_values(smhip, oracle):
    """A dense operand against a small one of at most 8 elements (per-channel mean / scale of interleaved data) through the
    LDS kernel: every element type, either operand order, RGB / RGBA / 8-channel, patterns that are broadcast in the
    middle, element counts with every tail.  (A register-resident variant of the kernel was tried for this shape and was
    slower: 66 us against 50 us for the RGB bias of tools/bcast_matrix.py.)"""
    cases = [((50, 60, 3), (3,)), ((7, 9, 11, 4), (1, 1, 1, 4)), ((1001, 3), (1, 3)), ((40, 50, 8), (8,)), ((6, 5, 100, 2), (6, 1, 1, 1)),
             ((3, 333, 2), (3, 1, 2)), ((20011, 5), (5,)), ((2, 4, 1025, 2), (2, 1, 1, 2))]
    for dtn, op in (("f32", "sub"), ("f64", "div"), ("i32", "mul"), ("i64", "add"), ("f32", "mul")):
        dt = DT[dtn]
        for big, small in cases:
            x = gen.gen(dt, int(np.prod(big)), 161, "uniform").reshape(big)
            y = gen.gen(dt, int(np.prod(small)), 162, "nonzero" if dtn[0] == "i" else "uniform").reshape(small)
            dx, dy = smhip.to_device(x), smhip.to_device(y)
            util.assert_same_bits(smhip.binary(sma.OPS[op], dx, dy).numpy(), oracle.binary(orc.OPS[op], x, y), f"{dtn} {op} {big} {small}")
            if not (dtn[0] == "i" and op == "div"):
                x2 = gen.gen(dt, int(np.prod(big)), 163, "nonzero" if dtn[0] == "i" else "uniform").reshape(big)
                util.assert_same_bits(smhip.binary(sma.OPS[op], dy, smhip.to_device(x2)).numpy(), oracle.binary(orc.OPS[op], y, x2),
                                      f"{dtn} {op} swapped {big} {small}")


def test_fewer_elements_than_a_vector(smhip, oracle):
    """1-3 output elements through every broadcast kernel's tail path (a launch must still have one workgroup)."""
    for dtn in ("f32", "f64", "i32", "i64"):
        dt = DT[dtn]
        base = gen.gen(dt, 64, 81, "uniform")
        other = gen.gen(dt, 64, 82, "uniform")
        db, do = smhip.to_device(base), smhip.to_device(other)
        for n in (1, 2, 3):
            for av, bv in ((base[:n], other[::2][:n]), (base[1::3][:n], other[:n]), (base[:n].reshape(n, 1), other[:n * 5:5].reshape(n, 1)),
                           (base[:n].reshape(1, n), other[3:4].reshape(1, 1)), (base[::7][:n], other[::5][:n])):
                got = smhip.binary(sma.OP_ADD, db.view_like(av, base), do.view_like(bv, other)).numpy()
                util.assert_same_bits(got, oracle.binary(orc.ADD, av, bv), f"{dtn} n={n} {av.strides} {bv.strides}")


def test_tile_kernel_vector_forms(smhip, oracle):
    """Extents and pitches that are multiples of the vector width, so the tile kernel's 16-byte form runs:
    one operand transposed (either side), both transposed (the Op is applied before the turn), partial
    patches on both axes, 4- and 8-byte elements."""
    for dtn, op in (("f32", "add"), ("f64", "sub"), ("i32", "mul"), ("i64", "add"), ("f64", "div")):
        dt = DT[dtn]
        for (r, c) in ((256, 192), (64, 128), (200, 72), (132, 516)):
            a = gen.gen(dt, r * c, 41, "uniform").reshape(c, r)
            b = gen.gen(dt, r * c, 42, "nonzero" if dtn[0] == "i" else "uniform").reshape(c, r)
            d = gen.gen(dt, r * c, 43, "nonzero" if dtn[0] == "i" else "uniform").reshape(r, c)
            da, db, dd = smhip.to_device(a), smhip.to_device(b), smhip.to_device(d)
            for (x, xb, dx), (y, yb, dy) in (((a.T, a, da), (b.T, b, db)), ((a.T, a, da), (d, d, dd)), ((d, d, dd), (b.T, b, db))):
                got = smhip.binary(sma.OPS[op], dx.view_like(x, xb), dy.view_like(y, yb)).numpy()
                util.assert_same_bits(got, oracle.binary(orc.OPS[op], x, y), f"{dtn} {op} {x.shape} {x.strides} {y.strides}")


def test_lds_kernel_patterns(smhip, oracle):
    """A dense operand against a small one that is broadcast along the inner axis or has a tiny inner extent
    (the LDS kernel): the reference tests' (N,224,224,3) op (1,224,1,3) family, either operand order, every
    dtype, element counts that are not multiples of the vector width, a staged operand above 4 KiB (persistent
    launch), six dimensions."""
    cases = [
        ((5, 224, 224, 3), (1, 224, 1, 3)),
        ((7, 31, 33, 3), (1, 1, 1, 3)),
        ((6, 100, 50, 4), (6, 1, 1, 4)),
        ((3, 5, 7, 11, 13, 3), (3, 1, 7, 1, 13, 1)),
        ((9, 40, 60, 5), (1, 40, 1, 5)),
        ((4, 300, 70, 6), (1, 300, 1, 6)),      # 1800 elements = 7.2 KiB staged (f32)
        ((33, 17, 5), (17, 1)),
    ]
    for dtn, op in (("f32", "add"), ("f64", "mul"), ("i32", "sub"), ("i64", "mul"), ("f32", "div")):
        dt = DT[dtn]
        for big, small in cases:
            x = gen.gen(dt, int(np.prod(big)), 51, "uniform").reshape(big)
            y = gen.gen(dt, int(np.prod(small)), 52, "nonzero" if dtn[0] == "i" else "uniform").reshape(small)
            dx, dy = smhip.to_device(x), smhip.to_device(y)
            util.assert_same_bits(smhip.binary(sma.OPS[op], dx, dy).numpy(), oracle.binary(orc.OPS[op], x, y), f"{dtn} {op} {big} {small}")
            if op != "div" or dtn[0] != "i":
                x2 = gen.gen(dt, int(np.prod(big)), 53, "nonzero" if dtn[0] == "i" else "uniform").reshape(big)
                util.assert_same_bits(smhip.binary(sma.OPS[op], dy, smhip.to_device(x2)).numpy(), oracle.binary(orc.OPS[op], y, x2),
                                      f"{dtn} {op} swapped {big} {small}")


def test_fused_equals_two_passes(smhip, oracle):
    """smhip_fused_contiguous: (a op1 b) op2 c in one pass is bit-identical to the two operator calls."""
    for dtn in ("f32", "f64", "i32", "i64"):
        for n in (1, 7, 1000, (1 << 20) + 3):
            a = gen.gen(DT[dtn], n, 51, "mixed")
            b = gen.gen(DT[dtn], n, 52, "nonzero" if dtn[0] == "i" else "mixed")
            c = gen.gen(DT[dtn], n, 53, "nonzero" if dtn[0] == "i" else "mixed")
            da, db, dc = smhip.to_device(a), smhip.to_device(b), smhip.to_device(c)
            for o1 in ("add", "sub", "mul", "div"):
                for o2 in ("add", "mul", "div", "sub"):
                    want = oracle.contiguous(orc.OPS[o2], oracle.contiguous(orc.OPS[o1], a, b), c)
                    util.assert_same_bits(smhip.fused(sma.OPS[o1], sma.OPS[o2], da, db, dc).numpy(), want, f"{dtn} {o1} {o2} {n}")
                    sv = 3 if dtn[0] == "i" else 1.5
                    want = oracle.array_scalar(orc.OPS[o2], oracle.contiguous(orc.OPS[o1], a, b), sv)
                    util.assert_same_bits(smhip.fused(sma.OPS[o1], sma.OPS[o2], da, db, sv).numpy(), want, f"{dtn} {o1} {o2} s {n}")
    with pytest.raises(sma.SmhipError):
        smhip.fused(sma.OP_POW, sma.OP_ADD, da, db, dc)


def test_complex_dot(smhip):
    """dot_product<std::complex<double>> (product.h:168-224): unconjugated sum a[i]*b[i]."""
    for n in (1, 2, 3, 511, 512, 513, 1000, 100003, 1024 * 512 + 77, 3_000_001):  # tile edges; the last two finish in several groups
        ar, ai = gen.gen(np.float64, n, 61, "uniform"), gen.gen(np.float64, n, 62, "uniform")
        br, bi = gen.gen(np.float64, n, 63, "uniform"), gen.gen(np.float64, n, 64, "uniform")
        a, b = ar + 1j * ai, br + 1j * bi
        da = smhip.to_device(a.view(np.float64))
        db = smhip.to_device(b.view(np.float64))
        got = smhip.dot_c64(da.ptr, db.ptr, n)
        want = complex(np.sum(a.astype(np.clongdouble) * b.astype(np.clongdouble)))
        scale = float(np.sum(np.abs(a) * np.abs(b)))
        assert abs(got - want) <= 4 * n * 2.0 ** -53 * scale + 1e-300, n
        # the asynchronous form leaves {re, im} in device memory: the same bits, run after run
        res = smhip.empty((2,), np.float64)
        smhip.dot_c64_async(da.ptr, db.ptr, n, res.ptr)
        first = res.numpy().copy()
        assert complex(first[0], first[1]) == got, n
        smhip.dot_c64_async(da.ptr, db.ptr, n, res.ptr)
        assert np.array_equal(res.numpy(), first), n


def test_complex_dot_against_the_oracle_and_the_recorded_reference(smhip, oracle):
    """the complex dot pinned to what the reference holds.  The oracle's `definition` (the scalar statement
    of product.h:221-222 for every element) is bit-identical to the compiled reference at n = 1 (tests/test_oracle.py); the
    GPU result must agree with it within the fp64 accumulation bound at every n, and at n = 1 -- one product, no
    accumulation -- within one rounding of each part (the kernel's fma chain rounds the two products of a part in the
    other order than GCC's contraction does).  The reference's own answers for n >= 2 (tests/golden/dot_extra.npz) count
    every paired product twice: the GPU result is checked to be what that doubling was applied to."""
    st = util.load_npz("dot_extra.npz")
    for c in cases.cdot_cases():
        a, b = cases.cdot_inputs(c)
        n = c["n"]
        da, db = smhip.to_device(a.view(np.float64)), smhip.to_device(b.view(np.float64))  # kept alive: the pool reuses freed blocks at once
        got = smhip.dot_c64(da.ptr, db.ptr, n)
        definition = oracle.dot_c64(a, b)
        scale = float(np.sum(np.abs(a) * np.abs(b)))
        assert abs(got - definition) <= 4 * n * 2.0 ** -53 * scale, n
        recorded = complex(*st[f"{c['id']}/out"])
        if n == 1:
            assert recorded == definition
            assert abs(got.real - recorded.real) <= 2.0 ** -52 * scale and abs(got.imag - recorded.imag) <= 2.0 ** -52 * scale
        else:
            paired = n - n % 2
            head = smhip.dot_c64(da.ptr, db.ptr, paired)  # the first `paired` elements of the same buffers
            tail = complex(a[-1] * b[-1]) if n % 2 else 0
            assert abs(recorded - (2 * head + tail)) <= 16 * n * 2.0 ** -53 * scale, n  # the AVX body's doubled sums, as data


def test_complex_float_dot(smhip, oracle):
    """The generic dot_product<T> with T = std::complex<float> (product.h:8-20).  The reference adds in float, sequentially;
    the kernel accumulates in fp64 and rounds once: it must sit within float rounding of the exact value and be no further
    from it than the reference's own recorded answer is (plus one rounding)."""
    st = util.load_npz("dot_extra.npz")
    rng = np.random.default_rng(17)
    extra = [(n, (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64),
              (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)) for n in (3, 1023, 1024, 1025, 1024 * 512 * 2 + 7)]
    todo = [(c["n"], *cases.cdot32_inputs(c), complex(*st[f"{c['id']}/out"])) for c in cases.cdot32_cases()] + [(n, a, b, None) for n, a, b in extra]
    for n, a, b, recorded in todo:
        da, db = smhip.to_device(a.view(np.float32)), smhip.to_device(b.view(np.float32))
        got = complex(smhip.dot_c32(da.ptr, db.ptr, n))
        exact = complex(np.sum(a.astype(np.complex128) * b.astype(np.complex128)))
        scale = float(np.sum(np.abs(a.astype(np.complex128)) * np.abs(b.astype(np.complex128))))
        ulp = float(np.spacing(np.float32(max(abs(exact.real), abs(exact.imag), 1e-30))))
        assert abs(got - exact) <= 2 * ulp + 4 * n * 2.0 ** -53 * scale, n          # fp64 accumulation, one rounding to float per part
        assert complex(oracle.dot_c32(a, b)) == pytest.approx(got, abs=2 * ulp), n   # the oracle's exact form
        if recorded is not None:
            assert abs(got - exact) <= abs(recorded - exact) + 2 * ulp, n            # never worse than the reference's float sums


def test_generic_integer_dot(smhip, oracle):
    """The generic dot_product<T> (product.h:8-20) for int8 / uint8 / int16 / uint16 / uint32 / uint64: bit-exact against
    the reference-recorded fixtures and the oracle -- the sum of products modulo 2^(8 sizeof T), in any order."""
    st = util.load_npz("dot_extra.npz")
    for c in cases.gdot_cases():
        a, b = cases.gdot_inputs(c)
        got = smhip.dot(smhip.to_device(a), smhip.to_device(b))
        assert got.dtype == a.dtype and got == st[f"{c['id']}/out"][0] == oracle.dot_int(a, b), c["id"]
    rng = np.random.default_rng(31)
    for dt in cases.GDOT_DTYPES:  # sizes that cross tiles and finish in several groups; unaligned starts
        info = np.iinfo(dt)
        for n in (0, 3, 4096 * 2 + 5, 1024 * 512 * (16 // np.dtype(dt).itemsize) + 77):
            a = rng.integers(info.min, info.max, size=n + 3, dtype=dt, endpoint=True)
            b = rng.integers(info.min, info.max, size=n + 3, dtype=dt, endpoint=True)
            da, db = smhip.to_device(a), smhip.to_device(b)
            for off in (0, 1, 3):
                va, vb = da.view_like(a[off:off + n], a), db.view_like(b[off:off + n], b)
                want = oracle.dot_int(np.ascontiguousarray(a[off:off + n]), np.ascontiguousarray(b[off:off + n])) if n else np.zeros(1, dt)[0]
                assert smhip.dot(va, vb) == want, (dt, n, off)


def test_user_op_via_hiprtc(smhip):
    """The plugin contract on the device: an Op registered as a HIP expression runs through the same
    entry points as the built-ins (contiguous, scalar, broadcast), for every element type."""
    op = smhip.register_op("(a + b) * 2")
    assert op >= 100 and smhip.register_op("(a + b) * 2") == op  # idempotent
    for dtn in ("f32", "f64", "i32", "i64"):
        for n in (1, 5, 4099):
            a = gen.gen(DT[dtn], n, 71, "uniform")
            b = gen.gen(DT[dtn], n, 72, "uniform")
            got = smhip.contiguous(op, smhip.to_device(a), smhip.to_device(b)).numpy()
            assert np.array_equal(got, ((a + b) * DT[dtn](2)).astype(DT[dtn])), (dtn, n)
        s = smhip.array_scalar(op, smhip.to_device(a), 3).numpy()
        assert np.array_equal(s, ((a + DT[dtn](3)) * DT[dtn](2)).astype(DT[dtn]))
    A = gen.gen(np.float32, 37 * 64, 73, "uniform").reshape(37, 64)
    r = gen.gen(np.float32, 64, 74, "uniform").reshape(1, 64)
    got = smhip.binary(op, smhip.to_device(A), smhip.to_device(r)).numpy()
    assert np.array_equal(got, (A + r) * np.float32(2))
    dA = smhip.to_device(A)
    got = smhip.binary(op, dA.view_like(A.T, A), smhip.to_device(np.ascontiguousarray(A.T))).numpy()
    assert np.array_equal(got, (A.T + A.T) * np.float32(2))
    # a second op with intrinsics, and one that does not compile
    hyp = smhip.register_op("sqrtf((float)(a * a + b * b))")
    a = gen.gen(np.float32, 1000, 75, "uniform"); b = gen.gen(np.float32, 1000, 76, "uniform")
    got = smhip.contiguous(hyp, smhip.to_device(a), smhip.to_device(b)).numpy()
    assert orc.ulp_diff_f32(got, np.sqrt(a * a + b * b)).max() <= 1
    bad = smhip.register_op("a +* b")
    with pytest.raises(sma.SmhipError) as e:
        smhip.contiguous(bad, smhip.to_device(a), smhip.to_device(b))
    assert "does not compile" in str(e.value)


def test_user_op_runs_the_broadcast_kernels(smhip, oracle):
    """A registered expression takes the same row / LDS / tile / gather kernels as the built-in Ops (jit.hip compiles the
    variant plan_launch() picks): "a * b" must reproduce the built-in multiply bit for bit on shapes that reach each of
    them, for every element type, including ragged rows, shifted views and a staged operand on the left."""
    user = smhip.register_op("a * b")
    for dtn in ("f32", "f64", "i32", "i64"):
        dt = DT[dtn]
        big = gen.gen(dt, 70 * 132, 91, "uniform").reshape(70, 132)
        oth = gen.gen(dt, 70 * 132, 92, "uniform").reshape(70, 132)
        sq = gen.gen(dt, 128 * 128, 93, "uniform").reshape(128, 128)
        img = gen.gen(dt, 6 * 20 * 24 * 3, 94, "uniform").reshape(6, 20, 24, 3)
        small = gen.gen(dt, 20 * 3, 95, "uniform").reshape(1, 20, 1, 3)
        dbig, doth, dsq, dimg, dsmall = (smhip.to_device(x) for x in (big, oth, sq, img, small))
        pairs = [
            (big, big, dbig, oth[3:4, :], oth, doth),                 # row kernel, row-constant operand
            (big[:, :131], big, dbig, oth[:, 5:6], oth, doth),        # row kernel, per-row scalar, ragged rows
            (big[1:60, 1:130], big, dbig, oth[2:61, 2:131], oth, doth),  # row kernel, three streams, shifted bases
            (sq.T, sq, dsq, sq, sq, dsq),                             # tile kernel, one operand turned
            (sq.T, sq, dsq, sq.T, sq, dsq),                           # tile kernel, both turned
            (sq[:100, :60].T, sq, dsq, sq[:60, :100], sq, dsq),       # tile kernel, partial patches
            (img, img, dimg, small, small, dsmall),                   # LDS kernel
            (small, small, dsmall, img, img, dimg),                   # LDS kernel, staged operand on the left
            (big[:, ::2], big, dbig, oth[:, 1::2], oth, doth),        # gather, inner-strided
            (big[:, :3], big, dbig, oth[:, 7:8], oth, doth),          # gather, tiny inner extent
        ]
        for av, abase, da, bv, bbase, db in pairs:
            x, y = da.view_like(av, abase), db.view_like(bv, bbase)
            got = smhip.binary(user, x, y).numpy()
            util.assert_same_bits(got, smhip.binary(sma.OP_MUL, x, y).numpy(), f"{dtn} {av.shape} {av.strides} x {bv.shape} {bv.strides}")
            util.assert_same_bits(got, oracle.binary(orc.MUL, av, bv), f"{dtn} vs oracle {av.shape} x {bv.shape}")


def test_jit_disk_cache(tmp_path):
    """hipRTC code objects are cached on disk (SMHIP_JIT_CACHE): a second process finds them, computes the same values and
    compiles nothing; SMHIP_JIT_CACHE=off leaves no files."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, time; sys.path.insert(0, %r); import numpy as np, simplemath_amd as sma; lib = sma.load();"
            "op = lib.register_op('(a + b) * 3'); a = lib.to_device(np.arange(4096, dtype=np.float32).reshape(64, 64));"
            "r = lib.to_device(np.ones((1, 64), dtype=np.float32)); t = time.perf_counter();"
            "out = lib.binary(op, a, r).numpy(); c = lib.contiguous(op, a, a).numpy(); dt = time.perf_counter() - t;"
            "print(float(out.sum()), float(c.sum()), dt)" % root)
    def run(cache):
        env = dict(os.environ, SMHIP_JIT_CACHE=cache)
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        a, b, dt = r.stdout.split()
        return float(a), float(b), float(dt)
    cache = str(tmp_path / "jit")
    first = run(cache)
    files = sorted(os.listdir(cache))
    assert len(files) >= 2 and all(f.endswith(".hsaco") for f in files), files
    second = run(cache)
    assert sorted(os.listdir(cache)) == files            # nothing new was compiled
    assert first[:2] == second[:2] == (float((np.arange(4096) + 1).sum() * 3), float(np.arange(4096).sum() * 6))
    assert second[2] < first[2]                          # and the cached run did not pay for hipRTC
    off = str(tmp_path / "off")
    os.mkdir(off)
    env_run = run("off")
    assert env_run[:2] == first[:2] and os.listdir(off) == []


def test_fused_expression_equals_the_operator_chain(smhip, oracle):
    """smhip_fused_expr: a whole expression in one pass must give the operator chain's values bit for bit (every
    operation rounds separately: no contraction), for every element type, every tail length, up to eight operands."""
    for dtn in ("f32", "f64", "i32", "i64"):
        dt = DT[dtn]
        for n in (1, 3, 4099, 100001):
            xs = [gen.gen(dt, n, 120 + k, "nonzero" if dtn[0] == "i" else "uniform") for k in range(8)]
            ds = [smhip.to_device(x) for x in xs]
            got = smhip.fused_expr("(a0 + a1) * a2 - a3", *ds[:4]).numpy()
            want = oracle.contiguous(orc.SUB, oracle.contiguous(orc.MUL, oracle.contiguous(orc.ADD, xs[0], xs[1]), xs[2]), xs[3])
            util.assert_same_bits(got, want, f"{dtn} n={n} four operands")
            got = smhip.fused_expr("a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7", *ds).numpy()
            want = xs[0]
            for k in range(1, 8):
                want = oracle.contiguous(orc.ADD, want, xs[k])
            util.assert_same_bits(got, want, f"{dtn} n={n} eight operands")
            got = smhip.fused_expr("a0 * a0", ds[0]).numpy()
            util.assert_same_bits(got, oracle.contiguous(orc.MUL, xs[0], xs[0]), f"{dtn} n={n} one operand")
            for alpha in (3, -2):  # run-time scalars: passed at launch
                got = smhip.fused_expr("a0 * s0 + a1 - s1", ds[0], ds[1], scalars=(alpha, 1)).numpy()
                want = oracle.array_scalar(orc.SUB, oracle.contiguous(orc.ADD, oracle.array_scalar(orc.MUL, xs[0], dt(alpha)), xs[1]), dt(1))
                util.assert_same_bits(got, want, f"{dtn} n={n} scalars {alpha}")
            # the sum of an expression in the same pass: equals the sum of the stored results (same adds), and of the oracle's chain
            total, stored = smhip.fused_expr_sum("(a0 - a1) * a2", ds[0], ds[1], ds[2], store=True)
            chain = oracle.contiguous(orc.MUL, oracle.contiguous(orc.SUB, xs[0], xs[1]), xs[2])
            util.assert_same_bits(stored.numpy(), chain, f"{dtn} n={n} expr+sum stored")
            only = smhip.fused_expr_sum("(a0 - a1) * a2", ds[0], ds[1], ds[2])
            assert total == only
            if dtn[0] == "i":
                assert total == smhip.sum(stored)   # integer totals are exact: any order gives the same value
            else:
                assert abs(total - oracle.sum(chain)) <= n * 2.0 ** -50 * (float(np.abs(chain.astype(np.float64)).sum()) + 1.0)
    with pytest.raises(sma.SmhipError):
        smhip.fused_expr("a0 +* a1", ds[0], ds[1])


def test_left_op_gathers_views(smhip):
    """SMHIP_OP_LEFT (out = a): the dense copy of strided / broadcast views that contiguous() and repeat()
    are built from; bit-exact including NaN payloads (nothing is computed)."""
    for dtn in ("f32", "f64", "i32", "i64"):
        a = gen.gen(DT[dtn], 24 * 40, 41, "wide").reshape(24, 40)
        da = smhip.to_device(a)
        dummy = smhip.to_device(np.zeros(1, dtype=DT[dtn]))
        for view in (a.T, a[3:19, 5:37], a[:, ::1], np.broadcast_to(a[:, None, :], (24, 3, 40))):
            dv = sma.DeviceArray(smhip, da.base_ptr, DT[dtn], view.shape, [s // a.itemsize for s in view.strides],
                                 (view.__array_interface__["data"][0] - a.__array_interface__["data"][0]) // a.itemsize, da._owner)
            got = smhip.binary(sma.OP_LEFT, dv, dummy).numpy()
            u = {4: np.uint32, 8: np.uint64}[a.itemsize]
            assert np.array_equal(got.view(u), np.ascontiguousarray(view).view(u)), (dtn, view.shape)


def test_dense_copy_of_inner_strided_views(smhip):
    """contiguous() of a view that takes every 2nd / 3rd / 4th element along the inner axis (stepped slices, one channel
    of interleaved data): the deinterleave kernel for rows of >= 64 outputs, the gather otherwise; against numpy, bit for
    bit, including the last rows of the allocation (nothing may be read past the view's last element)."""
    for dtn in ("f32", "f64", "i32", "i64"):
        a = gen.gen(DT[dtn], 37 * 1000, 151, "wide").reshape(37, 1000)
        da = smhip.to_device(a)
        dummy = smhip.to_device(np.zeros(1, dtype=DT[dtn]))
        flat = a.reshape(-1)
        views = [a[:, ::2], a[:, 1::2], a[:, ::3], a[:, 2::3], a[:, ::4], a[:, 3::4], a[5:, 7:900:3], a[:, ::5], a[:, :100:2], a[36:, 1::2],
                 flat[::2], flat[1::3], flat[36999 - 4 * 300::4]]
        for view in views:
            dv = sma.DeviceArray(smhip, da.base_ptr, DT[dtn], view.shape, [st // a.itemsize for st in view.strides],
                                 (view.__array_interface__["data"][0] - a.__array_interface__["data"][0]) // a.itemsize, da._owner)
            got = smhip.binary(sma.OP_LEFT, dv, dummy).numpy()
            u = {4: np.uint32, 8: np.uint64}[a.itemsize]
            assert np.array_equal(got.view(u), np.ascontiguousarray(view).view(u)), (dtn, view.shape, view.strides)


def test_flat_repeat_kernel(smhip):
    """repeat(r) of a dense array -- the (N, r) view with strides (1, 0) -- has a kernel of its own for r < 16: every
    repeat count, element width and tail length against numpy.repeat, bit for bit."""
    for dtn in ("f32", "f64", "i32", "i64"):
        for n in (1, 2, 5, 1000, 4097):
            a = gen.gen(DT[dtn], n, 131, "wide")
            da = smhip.to_device(a)
            dummy = smhip.to_device(np.zeros(1, dtype=DT[dtn]))
            for r in (1, 2, 3, 4, 5, 7, 8, 15, 16, 33):
                dv = sma.DeviceArray(smhip, da.base_ptr, DT[dtn], (n, r), (1, 0), 0, da._owner)
                got = smhip.binary(sma.OP_LEFT, dv, dummy).numpy().reshape(-1)
                u = {4: np.uint32, 8: np.uint64}[a.itemsize]
                assert np.array_equal(got.view(u), np.repeat(a, r).view(u)), (dtn, n, r)


def test_short_rows_against_per_row_values(smhip, oracle):
    """(N, r) op (N, 1) with r < 16 and N too large for the LDS kernel: its own kernel; either operand order, every
    element type and Op, rows that straddle vectors and workgroups, element counts with every tail."""
    for dtn, op in (("f32", "div"), ("f64", "mul"), ("i32", "sub"), ("i64", "add"), ("f32", "pow")):
        dt = DT[dtn]
        for n, r in ((20000, 3), (9001, 5), (10007, 15), (12345, 2), (8193, 7), (5001, 17), (3000, 31), (2003, 63), (1001, 250), (70, 1023)):  # (from 16 up: rows that are not whole vectors)
            kind = "positive" if op == "pow" else "uniform"
            x = gen.gen(dt, n * r, 141, kind).reshape(n, r)
            y = gen.gen(dt, n, 142, "nonzero" if dtn[0] == "i" else kind).reshape(n, 1)
            dx, dy = smhip.to_device(x), smhip.to_device(y)
            for lhs, rhs, dl, dr in ((x, y, dx, dy), (y, x, dy, dx)):
                got = smhip.binary(sma.OPS[op], dl, dr).numpy()
                want = oracle.binary(orc.OPS[op], lhs, rhs)
                if op == "pow":
                    assert orc.ulp_diff_f32(got, want).max() <= POW_ULP, (n, r)
                else:
                    util.assert_same_bits(got, want, f"{dtn} {op} ({n},{r}) {'x op y' if lhs is x else 'y op x'}")


def test_1d_strided_is_walked_not_assumed_dense(smhip):
    """the reference reads any 1-D operand as dense (calculate.h:10);
    the HIP path honours the strides (checked against numpy, the reference being UB here)."""
    a = gen.gen(np.float32, 64, 1, "uniform").reshape(8, 8)
    b = gen.gen(np.float32, 8, 2, "uniform")
    da, db = smhip.to_device(a), smhip.to_device(b)
    col = a[:, 3]  # 1-D, stride 8
    got = smhip.binary(sma.OP_ADD, da.view_like(col, a), db).numpy()
    assert np.array_equal(got, col + b)
    one = gen.gen(np.float32, 1, 3, "uniform")
    got = smhip.binary(sma.OP_MUL, db, smhip.to_device(one)).numpy()  # [N] * [1]
    assert np.array_equal(got, b * one)


def test_sum_and_fused_vs_oracle(smhip, oracle):
    for dt in ("f32", "f64", "i32", "i64"):
        for n in (1, 5, 2048, 100003, (1 << 21) + 3):
            a = gen.gen(DT[dt], n, 400 + n, "uniform")
            b = gen.gen(DT[dt], n, 500 + n, "uniform")
            da, db = smhip.to_device(a), smhip.to_device(b)
            s = smhip.sum(da)
            exact = oracle.sum(a)
            scale = float(np.abs(a.astype(np.float64)).sum())
            assert abs(s - exact) <= 1e-15 * scale + 1e-300, (dt, n)  # fp64 accumulation, any order
            out = smhip.empty((n,), DT[dt])
            sp = smhip.alloc(8)
            smhip.contiguous_sum_async(sma.OP_ADD, da, db, out, sp)
            fused = smhip.read_f64(sp)
            smhip.free(sp)
            want_out, want_sum = oracle.contiguous_sum(orc.ADD, a, b)
            util.assert_same_bits(out.numpy(), want_out, f"fused out {dt} {n}")
            assert abs(fused - want_sum) <= 1e-15 * float(np.abs(want_out.astype(np.float64)).sum()) + 1e-300


def test_dot_beats_reference_saturation(smhip):
    """the reference's f32 dot of 2^28 ones returns 2^27 (lane accumulators
    saturate at 2^24 each).  The fp64-accumulating reduction returns the exact count."""
    n = 1 << 26
    ones = smhip.full((n,), 1.0, np.float32)
    assert float(smhip.dot(ones, ones)) == float(n)
    assert smhip.sum(ones) == float(n)


def test_uniform_fill_matches_checker(smhip, oracle):
    for n, seed, first in ((1000, 1, 0), (4097, 6, 1 << 33)):
        got = smhip.uniform_f32(n, seed, -1.0, 1.0, first=first).numpy()
        assert np.array_equal(got, oracle.uniform_f32(n, seed, -1.0, 1.0, first=first))


# ---------------------------------- 4. BASELINE.json's full sizes, by properties
def _slices(n, width=1 << 16):
    return [0, width, n // 3, n // 2 + 1024, n - width]


def test_config2_full_size_add(smhip, oracle):
    """config 2: 1-D f32 add, N = 2^28.  Inputs from the counter-based generator, so any
    slice can be regenerated on the host: slices bit-exact vs the oracle + a checksum of
    the whole output (sum(c) = sum(a) + sum(b) to fp64 accumulation error)."""
    n, w = 1 << 28, 1 << 16
    a = smhip.uniform_f32(n, 1, -1.0, 1.0)
    b = smhip.uniform_f32(n, 2, -1.0, 1.0)
    c = smhip.contiguous(sma.OP_ADD, a, b)
    host = np.empty(w, dtype=np.float32)
    for off in _slices(n, w):
        smhip.download(host, c.ptr + off * 4)
        ha = oracle.uniform_f32(w, 1, -1.0, 1.0, first=off)
        hb = oracle.uniform_f32(w, 2, -1.0, 1.0, first=off)
        util.assert_same_bits(host, oracle.contiguous(orc.ADD, ha, hb), f"slice @{off}")
    sa, sb, sc = smhip.sum(a), smhip.sum(b), smhip.sum(c)
    # each c[i] carries <= 2^-24 relative rounding of a value < 2: |sum c - (sum a + sum b)| <= n * 2^-24
    assert abs(sc - (sa + sb)) <= n * 2.0 ** -24
    # idempotence / linearity: (a + b) - b == a exactly where no rounding happened is not guaranteed,
    # but (a + b) - (a + b) == 0 everywhere is
    z = smhip.contiguous(sma.OP_SUB, c, c)
    assert smhip.sum(z) == 0.0


def test_config3_full_size_broadcast_mul(smhip, oracle):
    """config 3: (4096 x 4096) * (1 x 4096) f32."""
    rows = cols = 4096
    A = smhip.uniform_f32(rows * cols, 3, -1.0, 1.0)
    r = smhip.uniform_f32(cols, 4, -1.0, 1.0)
    A2 = sma.DeviceArray(smhip, A.base_ptr, np.float32, (rows, cols), (cols, 1), 0, A._owner)
    r2 = sma.DeviceArray(smhip, r.base_ptr, np.float32, (1, cols), (cols, 1), 0, r._owner)
    out = smhip.binary(sma.OP_MUL, A2, r2)
    assert out.shape == (rows, cols)
    hr = oracle.uniform_f32(cols, 4, -1.0, 1.0)
    host = np.empty((8, cols), dtype=np.float32)
    for row0 in (0, 1, 2047, 4088):
        smhip.download(host, out.ptr + row0 * cols * 4)
        ha = oracle.uniform_f32(8 * cols, 3, -1.0, 1.0, first=row0 * cols).reshape(8, cols)
        util.assert_same_bits(host, oracle.binary(orc.MUL, ha, hr.reshape(1, cols)), f"rows @{row0}")
    # commutativity: r * A is the same array bit for bit
    out2 = smhip.binary(sma.OP_MUL, r2, A2)
    diff = smhip.contiguous(sma.OP_SUB, out, out2)
    assert smhip.sum(diff) == 0.0 and smhip.dot(diff, diff) == 0.0


def test_config4_full_size_pow(smhip, oracle):
    """config 4: pow(a, 2.5), N = 2^26, a in (0.01, 100)."""
    n, w = 1 << 26, 1 << 16
    a = smhip.uniform_f32(n, 5, 0.01, 100.0)
    p = smhip.array_scalar(sma.OP_POW, a, np.float32(2.5))
    host = np.empty(w, dtype=np.float32)
    for off in _slices(n, w):
        smhip.download(host, p.ptr + off * 4)
        ha = oracle.uniform_f32(w, 5, 0.01, 100.0, first=off)
        exact = np.power(ha.astype(np.float64), 2.5).astype(np.float32)
        assert orc.ulp_diff_f32(host, exact).max() <= POW_ULP
        assert orc.ulp_diff_f32(host, oracle.array_scalar(orc.POW, ha, np.float32(2.5))).max() <= POW_ULP
    # round trip: (a^2.5)^0.4 == a to a few ULP (each pow <= 1 ULP; the outer one amplifies by 0.4)
    back = smhip.array_scalar(sma.OP_POW, p, np.float32(0.4))
    smhip.download(host, back.ptr)
    ha = oracle.uniform_f32(w, 5, 0.01, 100.0)
    assert orc.ulp_diff_f32(host, ha).max() <= 4


def test_config5_shard_add_sum(smhip, oracle):
    """config 5's per-GPU shard: fused c = a + b, s = sum(c) at 2^28 elements, inputs
    uniform[0,1) with seeds 6/7 and the shard's global offset."""
    n, w, shard = 1 << 28, 1 << 16, 3
    first = shard * n
    a = smhip.uniform_f32(n, 6, 0.0, 1.0, first=first)
    b = smhip.uniform_f32(n, 7, 0.0, 1.0, first=first)
    c = smhip.empty((n,), np.float32)
    sp = smhip.alloc(8)
    smhip.contiguous_sum_async(sma.OP_ADD, a, b, c, sp)
    s = smhip.read_f64(sp)
    smhip.free(sp)
    host = np.empty(w, dtype=np.float32)
    for off in _slices(n, w):
        smhip.download(host, c.ptr + off * 4)
        ha = oracle.uniform_f32(w, 6, 0.0, 1.0, first=first + off)
        hb = oracle.uniform_f32(w, 7, 0.0, 1.0, first=first + off)
        util.assert_same_bits(host, oracle.contiguous(orc.ADD, ha, hb), f"slice @{off}")
    assert s == smhip.sum(c)  # the fused sum is the sum of what was stored (same order, same adds)
    assert abs(s - n) < 6 * np.sqrt(n / 6.0)  # E[a+b] = 1, var = 1/6: a 6-sigma sanity band


def test_many_short_rows(smhip, oracle):
    """2^22 rows of 64 floats against a (1 x 64) row: more workgroups along the row axis than a grid's y
    dimension (65 535) allows -- the row kernel's launch is 1-D."""
    rows, cols = 1 << 22, 64
    A = smhip.uniform_f32(rows * cols, 81, -1.0, 1.0)
    r = smhip.uniform_f32(cols, 82, -1.0, 1.0)
    A2 = sma.DeviceArray(smhip, A.base_ptr, np.float32, (rows, cols), (cols, 1), 0, A._owner)
    r2 = sma.DeviceArray(smhip, r.base_ptr, np.float32, (1, cols), (cols, 1), 0, r._owner)
    out = smhip.binary(sma.OP_MUL, A2, r2)
    hr = oracle.uniform_f32(cols, 82, -1.0, 1.0).reshape(1, cols)
    host = np.empty((1024, cols), dtype=np.float32)
    for row0 in (0, 70000 * 64 // 64, rows // 2 + 17, rows - 1024):
        smhip.download(host, out.ptr + row0 * cols * 4)
        ha = oracle.uniform_f32(1024 * cols, 81, -1.0, 1.0, first=row0 * cols).reshape(1024, cols)
        util.assert_same_bits(host, oracle.binary(orc.MUL, ha, hr), f"rows @{row0}")


def test_unsharded_config5_size_2p31(smhip, oracle):
    """N = 2^31 + 3 elements in ONE array (8 GiB per operand on a 288 GB device): global indices pass
    2^31 -- 64-bit indexing end to end (SURVEY section 7 "hard parts"), an odd tail, and the fused sum."""
    n, w = (1 << 31) + 3, 1 << 16
    a = smhip.uniform_f32(n, 6, 0.0, 1.0)
    b = smhip.uniform_f32(n, 7, 0.0, 1.0)
    c = smhip.empty((n,), np.float32)
    sp = smhip.alloc(8)
    smhip.contiguous_sum_async(sma.OP_ADD, a, b, c, sp)
    s = smhip.read_f64(sp)
    smhip.free(sp)
    host = np.empty(w, dtype=np.float32)
    for off in (0, 1 << 30, (1 << 31) - w - 5, n - w):  # up to and across the 2^31 boundary, incl. the 3-element tail
        smhip.download(host, c.ptr + off * 4)
        ha = oracle.uniform_f32(w, 6, 0.0, 1.0, first=off)
        hb = oracle.uniform_f32(w, 7, 0.0, 1.0, first=off)
        util.assert_same_bits(host, oracle.contiguous(orc.ADD, ha, hb), f"slice @{off}")
    # the fused sum and the plain sum of the stored result add the same 2^31 + 3 numbers in different groupings (one / two
    # vectors per lane, hence 2^21 / 2^20 workgroup partials): every partial is exact (the terms are multiples of 2^-24),
    # only the last few additions near 2^31 round, and they may round differently -- one ulp at most
    assert abs(s - smhip.sum(c)) <= 2 * np.spacing(s)
    assert abs(s - n) < 6 * np.sqrt(n / 6.0)
    d = smhip.contiguous(sma.OP_SUB, c, a)  # (a + b) - a: plain kernel at the same size
    smhip.download(host, d.ptr + (n - w) * 4)
    ha = oracle.uniform_f32(w, 6, 0.0, 1.0, first=n - w)
    hb = oracle.uniform_f32(w, 7, 0.0, 1.0, first=n - w)
    util.assert_same_bits(host, oracle.contiguous(orc.SUB, oracle.contiguous(orc.ADD, ha, hb), ha), "tail of (a+b)-a")
    del a, b, c, d
    smhip.pool_trim()


def test_results_do_not_depend_on_the_piece_size(tmp_path):
    """Very large operands go out as several launches (DESIGN.md section 3 'Very large arrays'): contiguous, scalar, pow, user-Op
    and reduction forms.  With SMHIP_PIECE_LOG2VEC=14 the same piece loops run at test sizes (3-5 pieces, whole and ragged
    last pieces, odd tails); every result -- the reductions' bits included -- must equal the single-launch result."""
    import subprocess, sys, os
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "piece_probe.py")
    files = {}
    for name, env in (("whole", {}), ("pieces", {"SMHIP_PIECE_LOG2VEC": "14"})):
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, probe, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0 and "piece_probe ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        files[name] = np.load(out)
    assert sorted(files["whole"].files) == sorted(files["pieces"].files) and len(files["whole"].files) >= 60
    for k in files["whole"].files:
        util.assert_same_bits(files["pieces"][k], files["whole"][k], k)
    # and the single-launch results are the right ones (the rest of this file pins them; one spot check here)
    n = (1 << 14) * 4 * 3 + 5
    w = files["whole"]
    assert np.array_equal(w[f"n{n}/fused_out"], w[f"n{n}/add"]) and abs(w[f"n{n}/fused_sum"][0] - w[f"n{n}/add"].astype(np.float64).sum()) < 1e-6 * n


def test_broadcast_forms_past_2p31(smhip):
    """Broadcast problems of more than 2^31 elements: the library cuts them along the outermost dimension into
    launches of < 2^31 (row kernel, LDS-tile kernel for a transposed operand, gather kernel for a stride-2 view,
    and a hipRTC user Op).  Blocks of rows around the cut (row 32764) and at the end are recomputed on the host from
    the downloaded operands; add / mul are one IEEE operation, so numpy is bit-exact."""
    R, Cc = (1 << 15) + 8, 1 << 16  # out: R x Cc = 2^31 + 2^19 elements
    n = R * Cc
    rows = (0, 32760, R - 8)

    def block(dev, pitch, row0, nrows):  # rows [row0, row0+nrows) of a dense (?, pitch) device matrix
        h = np.empty((nrows, pitch), dtype=np.float32)
        smhip.download(h, dev.base_ptr + row0 * pitch * 4)
        return h

    B = smhip.uniform_f32(n, 91, -1.0, 1.0)
    r = smhip.uniform_f32(Cc, 92, -1.0, 1.0)
    B2 = sma.DeviceArray(smhip, B.base_ptr, np.float32, (R, Cc), (Cc, 1), 0, B._owner)
    r2 = sma.DeviceArray(smhip, r.base_ptr, np.float32, (1, Cc), (Cc, 1), 0, r._owner)
    hr = r.numpy().reshape(1, Cc)
    # 1. row kernel
    out = smhip.binary(sma.OP_MUL, B2, r2)
    assert out.shape == (R, Cc)
    for row0 in rows:
        util.assert_same_bits(block(out, Cc, row0, 8), block(B, Cc, row0, 8) * hr, f"row kernel rows @{row0}")
    del out
    # 2. tile kernel: A stored (Cc, R), used transposed
    A = smhip.uniform_f32(n, 93, -1.0, 1.0)
    At = sma.DeviceArray(smhip, A.base_ptr, np.float32, (R, Cc), (1, R), 0, A._owner)
    out = smhip.binary(sma.OP_ADD, At, B2)
    c0 = 4096
    hA = block(A, R, c0, 64)  # A_store[c0:c0+64, :]
    for row0 in rows:
        got = block(out, Cc, row0, 8)[:, c0:c0 + 64]
        want = hA[:, row0:row0 + 8].T + block(B, Cc, row0, 8)[:, c0:c0 + 64]
        util.assert_same_bits(got, want, f"tile kernel rows @{row0}")
    del out, At
    # 3. gather kernel: every second element of A's first n (as an (R, Cc/2) matrix of stride-2 elements), times a column
    half = Cc // 2
    Av = sma.DeviceArray(smhip, A.base_ptr, np.float32, (R, half, 2), (Cc, 2, 0), 1, A._owner)   # value repeated twice
    Bv = sma.DeviceArray(smhip, B.base_ptr, np.float32, (R, half, 2), (Cc, 2, 1), 0, B._owner)   # B itself, dense
    out = smhip.binary(sma.OP_MUL, Av, Bv)
    assert out.size == n
    user = smhip.register_op("a * b")
    out_u = smhip.binary(user, Av, Bv)
    for row0 in rows:
        ha = block(A, Cc, row0, 8)
        want = np.repeat(ha[:, 1::2], 2, axis=1) * block(B, Cc, row0, 8)
        util.assert_same_bits(block(out, Cc, row0, 8), want, f"gather kernel rows @{row0}")
        util.assert_same_bits(block(out_u, Cc, row0, 8), want, f"user-op gather rows @{row0}")
    del out, out_u, A, B, Av, Bv, B2
    smhip.pool_trim()


# ------------------------------------------------------------------ 5. edge cases
def test_empty_and_tiny(smhip):
    e = smhip.empty((0,), np.float32)
    assert smhip.contiguous(sma.OP_ADD, e, e).size == 0
    assert smhip.sum(e) == 0.0
    z = smhip.empty((3, 0, 2), np.float32)
    assert smhip.binary(sma.OP_MUL, z, z).shape == (3, 0, 2)
    one = smhip.to_device(np.array([2.0], dtype=np.float32))
    assert smhip.binary(sma.OP_POW, one, one).numpy()[0] == 4.0


def test_misaligned_views(smhip, oracle):
    """Operands that start off a 16-byte boundary: the same vector kernels, with element-aligned 16-byte accesses."""
    n = 5000
    a = gen.gen(np.float32, n + 8, 11, "mixed")
    b = gen.gen(np.float32, n + 8, 12, "mixed")
    da, db = smhip.to_device(a), smhip.to_device(b)
    for oa, ob in ((1, 0), (0, 3), (1, 1), (2, 3)):
        va, vb = a[oa:oa + n], b[ob:ob + n]
        got = smhip.binary(sma.OP_ADD, da.view_like(va, a), db.view_like(vb, b)).numpy()
        util.assert_same_bits(got, oracle.contiguous(orc.ADD, np.ascontiguousarray(va), np.ascontiguousarray(vb)))
    u = gen.gen(np.float32, n + 8, 14, "uniform")
    du = smhip.to_device(u)
    assert abs(smhip.sum(du.view_like(u[3:3 + n], u)) - oracle.sum(np.ascontiguousarray(u[3:3 + n]))) < 1e-9
    assert abs(float(smhip.dot(du.view_like(u[1:1 + n], u), du.view_like(u[2:2 + n], u)))
               - float(np.dot(u[1:1 + n].astype(np.float64), u[2:2 + n].astype(np.float64)))) < 1e-2
    m = gen.gen(np.float32, 33 * 35, 13, "uniform").reshape(33, 35)
    dm = smhip.to_device(m)
    sub = m[1:, 1:]  # rows of 34 starting at odd offsets
    got = smhip.binary(sma.OP_SUB, dm.view_like(sub, m), dm.view_like(sub, m)).numpy()
    assert not got.any()


def test_ragged_and_shifted_rows(smhip, oracle):
    """Row kernel on 2-D / 3-D views whose row extent is not a multiple of the vector width, whose rows start at any
    element offset and whose pitches are odd: every dtype, dense x dense, dense x row, dense x column, scalar fill,
    fused and array-scalar forms on shifted bases."""
    for dtn, op in (("f32", "add"), ("f64", "mul"), ("i32", "sub"), ("i64", "add")):
        dt = DT[dtn]
        big = gen.gen(dt, 71 * 203, 61, "uniform").reshape(71, 203)
        oth = gen.gen(dt, 71 * 203, 62, "uniform").reshape(71, 203)
        dbig, doth = smhip.to_device(big), smhip.to_device(oth)
        for r0, r1, c0, c1 in ((0, 71, 0, 203), (1, 70, 1, 202), (3, 40, 5, 70), (0, 71, 2, 19), (2, 66, 7, 200), (0, 64, 3, 131)):
            va, vb = big[r0:r1, c0:c1], oth[r0:r1, c0:c1]
            got = smhip.binary(sma.OPS[op], dbig.view_like(va, big), doth.view_like(vb, oth)).numpy()
            util.assert_same_bits(got, oracle.binary(orc.OPS[op], va, vb), f"{dtn} dense x dense [{r0}:{r1},{c0}:{c1}]")
            row, col = oth[r0:r0 + 1, c0:c1], oth[r0:r1, c0:c0 + 1]
            got = smhip.binary(sma.OPS[op], dbig.view_like(va, big), doth.view_like(row, oth)).numpy()
            util.assert_same_bits(got, oracle.binary(orc.OPS[op], va, row), f"{dtn} dense x row [{r0}:{r1},{c0}:{c1}]")
            got = smhip.binary(sma.OPS[op], doth.view_like(col, oth), dbig.view_like(va, big)).numpy()
            util.assert_same_bits(got, oracle.binary(orc.OPS[op], col, va), f"{dtn} column x dense [{r0}:{r1},{c0}:{c1}]")
        for va, vb in ((big[:, 0:202:2], oth[:, 1::2]), (big[::3, 1:202:3], oth[::3, 2::3]), (big[:, ::2], oth[:, :102]), (big[5, ::7][:24], oth[::3, 0])):
            got = smhip.binary(sma.OPS[op], dbig.view_like(va, big), doth.view_like(vb, oth)).numpy()
            util.assert_same_bits(got, oracle.binary(orc.OPS[op], va, vb), f"{dtn} strided inner {va.shape} {va.strides} {vb.strides}")
        cube = gen.gen(dt, 9 * 11 * 37, 63, "uniform").reshape(9, 11, 37)
        dcube = smhip.to_device(cube)
        v = cube[1:8, 2:9, 3:36]
        w = cube[2:3, 2:9, 1:34]
        got = smhip.binary(sma.OPS[op], dcube.view_like(v, cube), dcube.view_like(w, cube)).numpy()
        util.assert_same_bits(got, oracle.binary(orc.OPS[op], v, w), f"{dtn} 3-D shifted views")
        # 1-D forms on bases shifted by 1..3 elements, lengths with every tail
        flat = gen.gen(dt, 4100, 64, "uniform")
        dflat = smhip.to_device(flat)
        for off, n in ((1, 4096), (3, 4093), (2, 7), (1, 1), (3, 1023)):
            x, y = flat[off:off + n], flat[4100 - off - n:4100 - off]
            dx, dy = dflat.view_like(x, flat), dflat.view_like(y, flat)
            util.assert_same_bits(smhip.binary(sma.OPS[op], dx, dy).numpy(), oracle.contiguous(orc.OPS[op], np.ascontiguousarray(x), np.ascontiguousarray(y)), f"{dtn} 1-D off {off} n {n}")
            sc = dt(3)
            util.assert_same_bits(smhip.array_scalar(sma.OPS[op], dx, sc).numpy(), oracle.array_scalar(orc.OPS[op], np.ascontiguousarray(x), sc), f"{dtn} scalar off {off} n {n}")


def test_strided_copy_assignment(smhip):
    """smhip_copy_strided (SMArray's `view = array`): numpy's own assignment is the specification.  Dense, sliced,
    transposed and broadcast sources into dense, sliced and transposed destinations; ragged rows; every element width."""
    for dt in (np.float32, np.float64, np.int32, np.int64):
        base = gen.gen(dt, 40 * 50 * 6, 71, "uniform").reshape(40, 50, 6)
        src = gen.gen(dt, 40 * 50 * 6, 72, "uniform").reshape(40, 50, 6)
        cases = [
            (np.s_[:, :, :], np.s_[:, :, :], None),
            (np.s_[3:30, 5:41, :], np.s_[0:27, 2:38, :], None),
            (np.s_[:, :, 2], np.s_[:, :, 5], None),               # inner-strided both sides
            (np.s_[1:38, 7, 1:6], np.s_[0:37, 3:8, 0], None),     # 2-D views of different pitch
            (np.s_[5, :, :], np.s_[7, :, :], None),
            (np.s_[2:35, 3:44, 1:4], None, (1, 41, 3)),           # broadcast source along the outer axis
        ]
        for dsel, ssel, bshape in cases:
            want = base.copy()
            if ssel is not None:
                sv = src[ssel]
            else:
                sv = src[:bshape[0], :bshape[1], :bshape[2]]
            want[dsel] = sv
            dbase, dsrc = smhip.to_device(base), smhip.to_device(src)
            smhip.assign(dbase.view_like(base[dsel], base), dsrc.view_like(sv, src))
            assert np.array_equal(dbase.numpy(), want), f"{np.dtype(dt)} {dsel} <- {ssel}"
        # transposed destination
        sq = gen.gen(dt, 64 * 80, 73, "uniform").reshape(64, 80)
        other = gen.gen(dt, 64 * 80, 74, "uniform").reshape(80, 64)
        want = sq.copy(); want.T[...] = other
        dsq = smhip.to_device(sq)
        smhip.assign(dsq.view_like(sq.T, sq), smhip.to_device(other))
        assert np.array_equal(dsq.numpy(), want)
    with pytest.raises(sma.SmhipError):
        z = smhip.to_device(np.zeros(8, np.float32))
        smhip._ck(smhip.c.smhip_copy_strided(0, C.c_void_p(z.ptr), (C.c_int64 * 1)(1), C.c_void_p(z.ptr), (C.c_int64 * 1)(0), (C.c_int64 * 1)(8), 1))


def test_fuzz_views_smoke(smhip):
    """A short run of tests/fuzz_views.py (random sliced / stepped / permuted / broadcast views of rank 1-5 against the
    oracle, random strided assignments against numpy) and of tests/fuzz_flat.py (1-D entry points, every tail and
    misalignment); 4 x 1500 and 2 x 800 cases of them ran clean when the kernels last changed."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_views.py"), "300", "17"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_flat.py"), "120", "17"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_records.py"), "60", "17"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_keep_store_is_followed_by_its_wait_states(smhip):
    """Regression: the `sc1` store is an inline asm, and gfx940-class hardware needs two wait states between a 16-byte
    store and a VALU write of its data registers -- which the compiler cannot insert behind an asm.  The column form of
    the flat tile kernel computes two results into the same registers back to back; without the `s_nop 1` inside the asm
    every fourth element of (8192, 672) * (8192, 1) was wrong (found by tests/fuzz_policy.py, seed 52 case 56)."""
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64, np.int32):
        w = 16 // np.dtype(dt).itemsize
        for rows, cols in ((8192, 168 * w), (8192, 256 * w), (16384, 168 * w)):
            if dt == np.int32:
                a = rng.integers(-1000, 1000, rows * cols, dtype=dt); b = rng.integers(-1000, 1000, rows, dtype=dt)
            else:
                a = rng.uniform(0.25, 4.0, rows * cols).astype(dt); b = rng.uniform(0.25, 4.0, rows).astype(dt)
            da, db = smhip.to_device(a), smhip.to_device(b)
            A = a.reshape(rows, cols)
            dA = sma.DeviceArray(smhip, da.base_ptr, dt, (rows, cols), (cols, 1), 0, da._owner)
            for shape, strides, ref in (((rows, 1), (1, 1), b.reshape(rows, 1)), ((1, cols), (cols, 1), b[:cols].reshape(1, cols))):
                dB = sma.DeviceArray(smhip, db.base_ptr, dt, shape, strides, 0, db._owner)
                for op, fn in ((sma.OP_MUL, np.multiply), (sma.OP_ADD, np.add)):
                    assert np.array_equal(smhip.binary(op, dA, dB).numpy(), fn(A, ref)), (dt, rows, cols, shape, op)


def test_periodic_small_operand_is_written_out_once(smhip):
    """A dense operand against a small one that ignores the leading axes (the reference tests' (N,224,224,3) op
    (1,224,1,3), a per-channel bias) at >= 128 MiB: the small operand's period is materialised and the flat tile kernel
    takes rows of it -- both operand orders, a period that needs repeating to fill a vector, f64, pow; against numpy."""
    rng = np.random.default_rng(101)
    def dev(arr): return smhip.to_device(np.ascontiguousarray(arr))
    x = rng.uniform(0.5, 2.0, (224, 224, 224, 3)).astype(np.float32)             # 128.6 MiB
    y = rng.uniform(0.5, 2.0, (1, 224, 1, 3)).astype(np.float32)
    dx, dy = dev(x), dev(y)
    assert np.array_equal(smhip.binary(sma.OP_ADD, dx, dy).numpy(), x + y)
    assert np.array_equal(smhip.binary(sma.OP_SUB, dx, dy).numpy(), x - y)
    assert np.array_equal(smhip.binary(sma.OP_MUL, dy, dx).numpy(), y * x)       # small operand on the left, commutative
    assert np.array_equal(smhip.binary(sma.OP_SUB, dy, dx).numpy(), y - x)       # ... not commutative: the LDS kernel
    got = smhip.binary(sma.OP_POW, dx, dy).numpy()
    want = np.power(x.astype(np.float64), y.astype(np.float64)).astype(np.float32)
    assert orc.ulp_diff_f32(got.reshape(-1), want.reshape(-1)).max() <= POW_ULP
    c = rng.uniform(0.5, 2.0, (3,)).astype(np.float32)                           # period 3 -> repeated to 12
    assert np.array_equal(smhip.binary(sma.OP_DIV, dx, dev(c)).numpy(), x / c)
    xd = rng.uniform(0.5, 2.0, (170, 224, 224, 2)).astype(np.float64)            # 130 MiB of f64; period 224*224*2
    yd = rng.uniform(0.5, 2.0, (224, 224, 1)).astype(np.float64)
    assert np.array_equal(smhip.binary(sma.OP_MUL, dev(xd), dev(yd)).numpy(), xd * yd)
    xi = rng.integers(-1000, 1000, (4097 * 8, 1024), dtype=np.int32)              # 128 MiB against a plain row: the flat route as before
    ri = rng.integers(-1000, 1000, (1024,), dtype=np.int32)
    assert np.array_equal(smhip.binary(sma.OP_ADD, dev(xi), dev(ri)).numpy(), xi + ri)
    x5 = rng.uniform(0.5, 2.0, (17, 7, 300000)).astype(np.float32)               # 136 MiB; the small operand ignores only the outermost axis
    y5 = rng.uniform(0.5, 2.0, (7, 1)).astype(np.float32)                        # period 7 * 300000 = 8 MiB: too long, stays where it was
    assert np.array_equal(smhip.binary(sma.OP_ADD, dev(x5), dev(y5)).numpy(), x5 + y5)  # (since round 3: one value per row, below)
    # a small operand that is constant along the TRAILING axes -- a per-channel bias in NCHW, (B, C, H, W) op (1, C, 1, 1): its
    # values are written out as one per row of (B C, H W) and the flat tile kernel's column form takes it
    xn = rng.uniform(0.5, 2.0, (42, 256, 56, 56)).astype(np.float32)              # 128.6 MiB
    cb = rng.uniform(0.5, 2.0, (1, 256, 1, 1)).astype(np.float32)
    dxn, dcb = dev(xn), dev(cb)
    assert np.array_equal(smhip.binary(sma.OP_ADD, dxn, dcb).numpy(), xn + cb)
    assert np.array_equal(smhip.binary(sma.OP_DIV, dxn, dcb).numpy(), xn / cb)
    assert np.array_equal(smhip.binary(sma.OP_MUL, dcb, dxn).numpy(), cb * xn)    # on the left, commutative
    assert np.array_equal(smhip.binary(sma.OP_SUB, dcb, dxn).numpy(), cb - xn)    # on the left, not commutative: the row kernel as before
    got = smhip.binary(sma.OP_POW, dxn, dcb).numpy()
    want = np.power(xn.astype(np.float64), cb.astype(np.float64)).astype(np.float32)
    assert orc.ulp_diff_f32(got.reshape(-1), want.reshape(-1)).max() <= POW_ULP
    del xn, dxn
    xs = rng.uniform(0.5, 2.0, (3, 7, 5, 256, 1200)).astype(np.float64)            # 123 MiB... of f64: (3, 7, 5, 256, 1200) * 8 B = 246 MiB
    ys = rng.uniform(0.5, 2.0, (7, 1, 256, 1)).astype(np.float64)                  # ignores axes 0 and 2, constant along the last
    assert np.array_equal(smhip.binary(sma.OP_SUB, dev(xs), dev(ys)).numpy(), xs - ys)


def test_periodic_route_with_leftover_rows(smhip):
    """Rows of 31 against one row at >= 128 MiB: four periods make whole vectors, and a row count that is not a multiple of
    four leaves up to three rows to a launch of their own (until round 3 such counts sent the whole problem to the row
    kernel); both operand orders, against numpy."""
    rng = np.random.default_rng(5)
    for rows, c in ((1083000 + 3, 31), (2000001, 17)):
        x = rng.uniform(0.5, 2.0, (rows, c)).astype(np.float32)
        r = rng.uniform(0.5, 2.0, (1, c)).astype(np.float32)
        dx, dr = smhip.to_device(x), smhip.to_device(r)
        assert np.array_equal(smhip.binary(sma.OP_SUB, dx, dr).numpy(), x - r), (rows, c)
        assert np.array_equal(smhip.binary(sma.OP_MUL, dr, dx).numpy(), r * x), (rows, c)


def test_fuzz_policy_smoke(smhip):
    """A short run of tests/fuzz_policy.py: random Ops on 4-70 MiB arrays, i.e. footprints on both sides of the stream-policy
    thresholds, through every kernel family that takes the policy word (3 x 160 cases ran clean when it last changed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_policy.py"), "20", "17"], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "ok: 20 policy cases" in r.stdout, r.stdout + r.stderr


def test_device_copy(smhip):
    """smhip_copy: word-sized copies run through the streaming kernel, anything else through the runtime's memcpy."""
    src = gen.gen(np.int32, 100003, 111, "wide")
    d = smhip.to_device(src)
    for nbytes, so, do in ((4 * 100003, 0, 0), (4 * 4099, 4 * 3, 4 * 1), (4, 8, 0), (1001, 0, 0), (4000, 2, 6)):
        out = smhip.to_device(np.zeros(100003, np.int32))
        smhip.copy(out.ptr + do, d.ptr + so, nbytes)
        want = np.zeros(100003, np.int32).view(np.uint8)
        want[do:do + nbytes] = src.view(np.uint8)[so:so + nbytes]
        assert np.array_equal(out.numpy().view(np.uint8), want), (nbytes, so, do)


def test_errors(smhip):
    a = smhip.to_device(np.zeros((2, 3), dtype=np.float32))
    b = smhip.to_device(np.zeros((4, 3), dtype=np.float32))
    with pytest.raises(RuntimeError, match="Cannot broadcast shapes"):
        smhip.binary(sma.OP_ADD, a, b)
    with pytest.raises(sma.SmhipError):
        smhip.elementwise_raw(sma.OP_ADD, np.float32, a.ptr, [1] * 7, a.ptr, [1] * 7, [1] * 7, a.ptr)
    with pytest.raises(sma.SmhipError):
        smhip.elementwise_raw(sma.OP_ADD, np.float32, a.ptr, [-1], a.ptr, [1], [3], a.ptr)


def test_int_division_definitions(smhip):
    """Where the reference traps (SIGFPE) the kernels are defined: x/0 = 0, INT_MIN/-1 = INT_MIN."""
    a = np.array([7, -7, 7, -7, -2147483648, 5, 0, -2147483648], dtype=np.int32)
    b = np.array([2, 2, -2, -2, -1, 0, 0, 1], dtype=np.int32)
    got = smhip.contiguous(sma.OP_DIV, smhip.to_device(a), smhip.to_device(b)).numpy()
    assert got.tolist() == [3, -3, -3, 3, -2147483648, 0, 0, -2147483648]


def test_pool_arenas_do_not_overlap(smhip):
    """Large blocks are carved out of shared slabs: random alloc / free traffic must never hand out
    overlapping ranges, extents must coalesce, and trimming must return empty slabs to the driver."""
    import random
    rnd = random.Random(7)
    smhip.synchronize()
    smhip.pool_trim()
    base_use, _ = smhip.pool_stats()
    live = {}  # ptr -> (nbytes, tag)
    tag = 0
    for step in range(300):
        if live and (rnd.random() < 0.45 or len(live) > 24):
            p = rnd.choice(list(live))
            nbytes, t = live.pop(p)
            host = np.empty(4, dtype=np.int32)
            smhip.download(host, p)
            assert (host == t).all(), "block was overwritten while live"
            smhip.download(host, p + nbytes - 16)
            assert (host == t).all(), "block tail was overwritten while live"
            smhip.free(p)
        else:
            nbytes = rnd.choice([1 << 20, 3 << 20, (5 << 20) + 4096, 17 << 20, 64 << 20, 100 << 20, 257 << 20, 1000, 70000])
            nbytes = (nbytes + 15) // 16 * 16
            p = smhip.alloc(nbytes)
            tag += 1
            for q, sz in live.items():
                assert p + nbytes <= q or q + sz[0] <= p, "overlapping allocations"
            v = np.array([tag], dtype=np.int32)
            smhip.c.smhip_fill(2, C.c_void_p(p), v.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes // 4))
            live[p] = (nbytes, tag)
    for p in list(live):
        smhip.free(p)
    smhip.synchronize()
    use, cached_before = smhip.pool_stats()
    assert use == base_use and cached_before > 0
    smhip.pool_trim()
    use, cached = smhip.pool_stats()
    # empty slabs went back to the driver; one that still holds a live block of an earlier test keeps its free remainder
    assert use == base_use and cached < cached_before and cached <= (1 << 30)
    # three same-sized operands come out of one slab, back to back (what keeps the stream's rate stable)
    a, b, c = smhip.alloc(64 << 20), smhip.alloc(64 << 20), smhip.alloc(64 << 20)
    assert b - a == 64 << 20 and c - b == 64 << 20
    for p in (a, b, c):
        smhip.free(p)


def test_pool_reuses_buffers(smhip):
    smhip.synchronize()
    p = smhip.alloc(1 << 20)
    smhip.free(p)
    q = smhip.alloc(1 << 20)
    assert p == q  # the per-operator `new T[n]` of SMArray.h:219 became a pointer pop
    smhip.free(q)


def test_inline_operands_vs_oracle(smhip, oracle):
    """smhip_elementwise_inline: tiny host-built operands carried by the launch packet (no upload).  Same values as the
    uploaded path / the oracle for every Op and element type, dense and broadcast shapes, array and scalar operands."""
    from oracle import oracle as orc
    rng = np.random.default_rng(11)
    ops = {"add": (sma.OP_ADD, orc.ADD), "sub": (sma.OP_SUB, orc.SUB), "mul": (sma.OP_MUL, orc.MUL), "div": (sma.OP_DIV, orc.DIV)}
    shapes = [((5, 5), (5, 5)), ((25,), (25,)), ((2, 3), (1, 3)), ((4, 1), (1, 7)), ((3, 1, 5), (2, 1)), ((1,), (13,)), ((2, 3, 2, 2), (3, 1, 2))]
    for dt in (np.float32, np.float64, np.int32, np.int64):
        for sha, shb in shapes:
            if np.issubdtype(dt, np.floating):
                a, b = rng.uniform(-4, 4, sha).astype(dt), rng.uniform(0.5, 4, shb).astype(dt)
            else:
                a, b = rng.integers(-50, 50, sha).astype(dt), rng.integers(1, 9, shb).astype(dt)
            for name, (op, oop) in ops.items():
                want = oracle.binary(oop, a, b)
                got = smhip.binary_inline(op, a, b).numpy()
                assert np.array_equal(got, want), (dt, sha, shb, name)
                da = smhip.to_device(a)                      # one side resident, the other inline
                assert np.array_equal(smhip.binary_inline(op, da, b).numpy(), want), (dt, sha, shb, name, "a resident")
                db = smhip.to_device(b)
                assert np.array_equal(smhip.binary_inline(op, a, db).numpy(), want), (dt, sha, shb, name, "b resident")
            s = dt(3)
            assert np.array_equal(smhip.binary_inline(sma.OP_MUL, a, s).numpy(), (a * s).astype(dt)), (dt, sha, "scalar")
    # int pow: the reference's square-and-multiply (crafted_pow.h:54-103), as the array kernels compute it
    base = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], dtype=np.int32)
    assert np.array_equal(smhip.binary_inline(sma.OP_POW, base, np.int32(3)).numpy(), base ** 3)
    x = rng.uniform(0.01, 100, 200).astype(np.float32)
    got = smhip.binary_inline(sma.OP_POW, x, np.float32(2.5)).numpy()
    assert orc.ulp_diff_f32(got, np.power(x.astype(np.float64), 2.5).astype(np.float32)).max() <= 4
    # limits are refusals, not truncations
    with pytest.raises(sma.SmhipError):
        smhip.binary_inline(sma.OP_ADD, np.zeros(257, np.float32), np.zeros(257, np.float32))  # 1028 bytes
    with pytest.raises(sma.SmhipError):
        smhip.binary_inline(sma.OP_ADD, np.zeros((100, 1), np.float32), np.zeros((1, 100), np.float32))  # 10 000 results


def test_inner_strided_operands_take_wide_loads(smhip, oracle):
    """a[:, ::S] op b[...] for S = 2, 3, 4 against a dense / per-row / equally strided partner (strided_row_kernel: S wide loads
    + select instead of one 4-byte load per lane): every element type, ragged row lengths, views that end at the very
    last element of their allocation, 1-D and 3-D forms; bit-exact against the oracle's element_wise_op walk."""
    from oracle import oracle as orc
    for dtn in ("f32", "f64", "i32", "i64"):
        dt = DT[dtn]
        a = gen.gen(dt, 23 * 1030, 7, "uniform").reshape(23, 1030)
        b = gen.gen(dt, 23 * 1030, 8, "positive" if dtn[0] == "f" else "uniform").reshape(23, 1030)
        if dtn[0] == "i":
            b = np.where(b == 0, 1, b).astype(dt)
        da, db = smhip.to_device(a), smhip.to_device(b)
        fa, fb = a.reshape(-1), b.reshape(-1)
        pairs = [(a[:, ::2], b[:, ::2]), (a[:, 1::2], b[:, ::2]), (a[:, ::3], b[:, 1::3]), (a[:, ::4], b[:, 2::4]),
                 (a[:, ::2], b[:, :515]), (a[:, :343], b[:, 1::3]),                     # strided against dense
                 (a[:, ::4], b[:, 5:6]), (a[:, 3:4], b[:, ::2]),                        # strided against one value per row
                 (a[3:, 6:1030:2], b[3:, 7:1030:2]), (a[22:, ::2], b[22:, 1::2]),        # ends at the allocation's last element
                 (a[:, ::4], b[:, :258]), (a[:, :258], b[:, 1::4]), (a[:, 2::4], b[:, 3::4]),  # stride 4: rows that end inside / at a 256-output chunk (first of a pair: a view of a, second: of b)
                 (fa[::2], fb[1::2]), (fa[23 * 1030 - 4 * 900::4], fb[:900]),           # 1-D
                 (a.reshape(23, 10, 103)[:, :, ::2][:, :, :40], b.reshape(23, 10, 103)[:, :, 1::2][:, :, :40])]  # short rows: gather
        for k, (va, vb) in enumerate(pairs):
            if va.shape[-1] != vb.shape[-1] and 1 not in (va.shape[-1], vb.shape[-1]):  # stepped slices differ by one column
                w = min(va.shape[-1], vb.shape[-1])
                va, vb = va[..., :w], vb[..., :w]
            for opn in ("add", "div"):
                want = oracle.binary(getattr(orc, opn.upper()), va, vb)
                got = smhip.binary(sma.OPS[opn], da.view_like(va, a), db.view_like(vb, b)).numpy()
                assert np.array_equal(got, want), (dtn, k, opn, va.shape, va.strides, vb.strides)
    # a user-defined Op takes the same kernel through hipRTC
    op = smhip.register_op("a * b + a")
    a = gen.gen(np.float32, 64 * 512, 9, "uniform").reshape(64, 512)
    b = gen.gen(np.float32, 64 * 512, 10, "uniform").reshape(64, 512)
    da, db = smhip.to_device(a), smhip.to_device(b)
    got = smhip.binary(op, da.view_like(a[:, ::2], a), db.view_like(b[:, 1::2], b)).numpy()
    assert np.array_equal(got, a[:, ::2] * b[:, 1::2] + a[:, ::2])


def test_large_view_shapes_vs_numpy(smhip):
    """Views big enough for whole 64 x 128 LDS patches, long strided rows and multi-row lanes (tests/fuzz_views.py keeps its
    extents under 140): random 2-D / 3-D bases with extents up to ~1500, transposed / permuted / stepped / offset views of
    them, + and * in f32, f64 and i32, bit-exact against numpy (one correctly rounded operation per element)."""
    rng = np.random.default_rng(2024)
    for t in range(48):
        dtn = ("f32", "f64", "i32")[t % 3]
        dt = DT[dtn]
        nd = 2 if t % 4 else 3
        dims = [int(rng.integers(130, 1500)), int(rng.integers(130, 1500))] if nd == 2 else [int(rng.integers(2, 9)), int(rng.integers(70, 400)), int(rng.integers(130, 700))]
        a = gen.gen(dt, int(np.prod(dims)), 300 + t, "uniform").reshape(dims)
        b = gen.gen(dt, int(np.prod(dims)), 400 + t, "uniform").reshape(dims)

        step = int(rng.integers(2, 5))

        def view(x, kind):
            if kind == 0:
                return x
            if kind == 1:
                return np.transpose(x, list(range(x.ndim - 2)) + [x.ndim - 1, x.ndim - 2])
            if kind == 2:
                return x[..., ::step]
            if kind == 3:
                return x[..., 1:-1, 3:-2]
            return np.transpose(x)  # full reversal of the axes

        ka, kb = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        va, vb = view(a, ka), view(b, kb)
        if va.shape != vb.shape:  # make them broadcast-compatible: bring b to a's shape through a's own kind on b
            vb = view(b, ka)
        da, db = smhip.to_device(a), smhip.to_device(b)
        for opn, f in (("add", np.add), ("mul", np.multiply)):
            got = smhip.binary(sma.OPS[opn], da.view_like(va, a), db.view_like(vb, b)).numpy()
            assert np.array_equal(got, f(va, vb)), (t, dtn, opn, dims, ka, kb)


def test_small_planes_batched(smhip):
    """A dense batch of small planes with one operand read transposed -- (B, n, m) seen as (B, m, n) -- takes the planes
    kernel (broadcast.hip): plane shapes from 2 x 2 to 64 x 64 incl. odd ones, batch counts that end inside a workgroup's
    chunk, the transposed operand on either side of a non-commutative Op, every element type, plain copies; numpy is the
    specification."""
    rng = np.random.default_rng(99)
    t = 0
    for n, m in ((2, 2), (3, 3), (4, 4), (3, 5), (7, 2), (8, 8), (12, 12), (16, 16), (16, 3), (3, 16), (24, 24), (31, 33), (32, 32), (48, 48), (64, 64), (5, 100), (100, 5)):
        for B in (64, 1000, 4099):
            t += 1
            dtn = ("f32", "f64", "i32", "i64")[t % 4]
            dt = DT[dtn]
            x = gen.gen(dt, B * n * m, 100 + t, "uniform").reshape(B, n, m)
            y = gen.gen(dt, B * n * m, 200 + t, "uniform").reshape(B, m, n)
            dx, dy = smhip.to_device(x), smhip.to_device(y)
            xt = np.transpose(x, (0, 2, 1))
            got = smhip.binary(sma.OPS["sub"], dx.view_like(xt, x), dy).numpy()
            assert np.array_equal(got, xt - y), ("planes", n, m, B, dtn)
            got = smhip.binary(sma.OPS["sub"], dy, dx.view_like(xt, x)).numpy()
            assert np.array_equal(got, y - xt), ("planes swapped", n, m, B, dtn)
            dst = smhip.empty((B, m, n), dt)
            smhip.assign(dst, dx.view_like(xt, x))
            assert np.array_equal(dst.numpy(), xt), ("planes copy", n, m, B, dtn)
    for dt in (np.float32, np.float64):  # pow: the tables share the LDS with the tile
        base = rng.uniform(0.05, 30.0, (3000, 6, 10)).astype(dt)
        e = rng.uniform(-3.0, 3.0, (3000, 10, 6)).astype(dt)
        bt = np.transpose(base, (0, 2, 1))
        got = smhip.binary(sma.OP_POW, smhip.to_device(base).view_like(bt, base), smhip.to_device(e)).numpy()
        with np.errstate(all="ignore"):
            exact = np.power(bt.astype(np.longdouble), e.astype(np.longdouble)).astype(dt)
        it = np.int32 if dt == np.float32 else np.int64
        assert np.abs(got.view(it).astype(np.int64) - exact.view(it).astype(np.int64)).max() <= 1, dt


def test_few_long_rows_against_one_row(smhip):
    """Config 3's shape with FEW LONG rows -- (6, 2^21) op (1, 2^21): the flat tile kernel walks such arrays column block by
    column block (eight workgroups on one row's neighbouring tiles, the next eight on the next row's) so that the broadcast
    row stays in the L2; every tile must still be visited exactly once: both operand orders, cold and replayed operands
    (one / two vectors per lane), f32 / f64 / i32, against numpy."""
    rng = np.random.default_rng(77)
    for dtn, cols in (("f32", 1 << 21), ("f64", 1 << 20), ("i32", 3 << 20)):
        dt = DT[dtn]
        rows = 6 if dtn != "i32" else 2
        x = gen.gen(dt, rows * cols, 31, "uniform").reshape(rows, cols)
        r = gen.gen(dt, cols, 32, "uniform").reshape(1, cols)
        dx, dr = smhip.to_device(x), smhip.to_device(r)
        for rep in range(2):  # the second call finds its operands warm
            assert np.array_equal(smhip.binary(sma.OP_SUB, dx, dr).numpy(), x - r), (dtn, rep)
            assert np.array_equal(smhip.binary(sma.OP_MUL, dr, dx).numpy(), r * x), (dtn, rep)


def test_record_kernel_aos_soa(smhip):
    """Planes with one tiny extent and one turned operand (arrays of small records <-> few long rows) take the record kernel
    (broadcast.hip): records of 2 ... 128 elements, both directions, the turned operand on either side of a non-commutative
    Op, record counts that end inside a workgroup's chunk, padded row pitches, every element type, and the same shapes as
    plain copies (`dst = src.T`); numpy is the specification (one correctly rounded operation per element)."""
    rng = np.random.default_rng(4242)
    t = 0
    for k in (2, 3, 4, 5, 7, 8, 12, 13, 15, 16, 24, 32, 36, 48, 100, 128):
        for n in (4096, 5000 + k, 70001):
            t += 1
            dtn = ("f32", "f64", "i32", "i64")[t % 4]
            dt = DT[dtn]
            opn, f = (("sub", np.subtract), ("add", np.add), ("div", np.divide), ("mul", np.multiply))[t % 4] if dtn[0] == "f" else (("sub", np.subtract), ("mul", np.multiply))[t % 2]
            recs = gen.gen(dt, n * k, 500 + t, "positive" if opn == "div" else "uniform").reshape(n, k)      # n records of k
            rows = gen.gen(dt, n * k, 600 + t, "positive" if opn == "div" else "uniform").reshape(k, n)      # k rows of n
            drecs, drows = smhip.to_device(recs), smhip.to_device(rows)
            # AoS -> SoA: out (k, n) = recs.T op rows, and with the operands exchanged
            got = smhip.binary(sma.OPS[opn], drecs.view_like(recs.T, recs), drows).numpy()
            assert np.array_equal(got, f(recs.T, rows)), ("aos->soa", k, n, dtn, opn)
            got = smhip.binary(sma.OPS[opn], drows, drecs.view_like(recs.T, recs)).numpy()
            assert np.array_equal(got, f(rows, recs.T)), ("aos->soa swapped", k, n, dtn, opn)
            # SoA -> AoS: out (n, k) = rows.T op recs
            got = smhip.binary(sma.OPS[opn], drows.view_like(rows.T, rows), drecs).numpy()
            assert np.array_equal(got, f(rows.T, recs)), ("soa->aos", k, n, dtn, opn)
            got = smhip.binary(sma.OPS[opn], drecs, drows.view_like(rows.T, rows)).numpy()
            assert np.array_equal(got, f(recs, rows.T)), ("soa->aos swapped", k, n, dtn, opn)
            # against ONE long row for all k rows
            one = rows[:1]
            got = smhip.binary(sma.OPS[opn], drecs.view_like(recs.T, recs), drows.view_like(one, rows)).numpy()
            assert np.array_equal(got, f(recs.T, one)), ("aos->soa against one row", k, n, dtn, opn)
            # the k rows inside rows of a longer pitch (a slice of a wider array)
            wide = gen.gen(dt, k * (n + 37), 700 + t, "uniform").reshape(k, n + 37)
            dwide = smhip.to_device(wide)
            sl = wide[:, 5:5 + n]
            got = smhip.binary(sma.OPS["add"], dwide.view_like(sl.T, wide), drecs).numpy()
            assert np.array_equal(got, sl.T + recs), ("soa->aos pitch", k, n, dtn)
            got = smhip.binary(sma.OPS["add"], drecs.view_like(recs.T, recs), dwide.view_like(sl, wide)).numpy()
            assert np.array_equal(got, recs.T + sl), ("aos->soa pitch", k, n, dtn)
            # copies: dst = src.T both ways
            dst = smhip.empty((k, n), dt)
            smhip.assign(dst, drecs.view_like(recs.T, recs))
            assert np.array_equal(dst.numpy(), recs.T), ("copy aos->soa", k, n, dtn)
            dst2 = smhip.empty((n, k), dt)
            smhip.assign(dst2, drows.view_like(rows.T, rows))
            assert np.array_equal(dst2.numpy(), rows.T), ("copy soa->aos", k, n, dtn)
    # a batch of such planes: (B, n, k) viewed as (B, k, n) -- NHWC <-> NCHW of 224 x 224 x 3 images among them (one launch:
    # the workgroup index carries the plane; 50176 records go out as 28 whole chunks of 1792)
    for B, n, k, dtn in ((3, 90000, 3, "f32"), (2, 70000, 4, "f64"), (4, 66000, 12, "i32"), (5, 5000, 3, "f32"), (8, 50176, 3, "f32"), (3, 50180, 3, "i64")):
        dt = DT[dtn]
        recs = gen.gen(dt, B * n * k, 801, "uniform").reshape(B, n, k)
        rows = gen.gen(dt, B * n * k, 802, "uniform").reshape(B, k, n)
        drecs, drows = smhip.to_device(recs), smhip.to_device(rows)
        rt, wt = np.transpose(recs, (0, 2, 1)), np.transpose(rows, (0, 2, 1))
        got = smhip.binary(sma.OPS["sub"], drecs.view_like(rt, recs), drows).numpy()
        assert np.array_equal(got, rt - rows), ("batched aos->soa", B, n, k, dtn)
        got = smhip.binary(sma.OPS["sub"], drecs, drows.view_like(wt, rows)).numpy()
        assert np.array_equal(got, recs - wt), ("batched soa->aos", B, n, k, dtn)
        dst = smhip.empty((B, k, n), dt)
        smhip.assign(dst, drecs.view_like(rt, recs))
        assert np.array_equal(dst.numpy(), rt), ("batched copy", B, n, k, dtn)
    # integer pow (wrapping square-and-multiply, crafted_pow.h:54-103): non-negative exponents, numpy wraps the same way
    for dt in (np.int32, np.int64):
        base = rng.integers(-9, 10, (5000, 5)).astype(dt)
        e = rng.integers(0, 12, (5, 5000)).astype(dt)
        got = smhip.binary(sma.OP_POW, smhip.to_device(base).view_like(base.T, base), smhip.to_device(e)).numpy()
        with np.errstate(all="ignore"):
            assert np.array_equal(got, np.power(base.T, e)), dt
    # pow through it (tables in LDS next to the tile), float and double
    for dt in (np.float32, np.float64):
        base = rng.uniform(0.05, 30.0, (6000, 6)).astype(dt)
        e = rng.uniform(-3.0, 3.0, (6, 6000)).astype(dt)
        got = smhip.binary(sma.OP_POW, smhip.to_device(base).view_like(base.T, base), smhip.to_device(e)).numpy()
        with np.errstate(all="ignore"):
            exact = np.power(base.T.astype(np.longdouble), e.astype(np.longdouble)).astype(dt)
        it = np.int32 if dt == np.float32 else np.int64
        assert np.abs(got.view(it).astype(np.int64) - exact.view(it).astype(np.int64)).max() <= 1, dt


def test_tile_kernel_patch_shapes():
    """The tile kernel has three patch shapes (64 x 512 B on a diagonal walk up to 256 MiB per array, 64 x 1024 B row-major
    beyond, 64 x 128 B for planes whose q extent is a few dozen elements: DESIGN.md section 3).  tests/tile_probe.py runs
    transposed / permuted views of every element type, + * pow and a user Op against numpy with each shape forced at test
    sizes; unforced, its skinny cases take the short patch and its 272 MiB case the wide one."""
    import subprocess, sys, os
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_probe.py")
    for env, args in (({"SMHIP_TILE_QB": "1024"}, []), ({"SMHIP_TILE_QB": "512"}, []), ({"SMHIP_TILE_QB": "128"}, []), ({}, ["big"])):
        clean = {k: v for k, v in os.environ.items() if k not in ("SMHIP_TILE_WIDE", "SMHIP_TILE_QB")}
        r = subprocess.run([sys.executable, probe] + args, capture_output=True, text=True, timeout=600, env=dict(clean, **env))
        assert r.returncode == 0 and "tile_probe ok" in r.stdout, (env, r.stdout[-2000:] + r.stderr[-2000:])


def test_user_ops_first_used_from_many_threads(smhip, tmp_path, monkeypatch):
    """hipRTC builds run with the cache mutex released: threads that need the SAME new kernel wait for one build, threads
    that need different ones compile side by side -- and every thread gets the right kernel."""
    import threading
    exprs = ["a * b + (T)%d" % k for k in range(101, 104)]   # fresh expressions: nothing cached in this process
    ids = [smhip.register_op(e) for e in exprs]
    n = 100003
    a = gen.gen(np.float32, n, 1, "uniform"); b = gen.gen(np.float32, n, 2, "uniform")
    da, db = smhip.to_device(a), smhip.to_device(b)
    errors = []

    def work(k):
        try:
            for rep in range(3):
                got = smhip.contiguous(ids[k % 3], da, db).numpy()
                want = a * b + np.float32(101 + k % 3)
                if not np.array_equal(got, want):
                    errors.append((k, rep, "values"))
                v = smhip.binary(ids[k % 3], da.view_like(a[: n - 3].reshape(-1, 100)[:, ::2], a), db.view_like(b[: n - 3].reshape(-1, 100)[:, 1::2], b)).numpy()
                if not np.array_equal(v, a[: n - 3].reshape(-1, 100)[:, ::2] * b[: n - 3].reshape(-1, 100)[:, 1::2] + np.float32(101 + k % 3)):
                    errors.append((k, rep, "broadcast values"))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(9)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_store_policy_keep_arm(smhip):
    """Results whose launch footprint (reads + writes) lies between 40 and 256 MiB are stored with `sc1` instead of
    non-temporally (csrc/internal.h: stream_policy) -- a different store instruction in every streaming kernel.  One case
    per kernel, sized into that window, against numpy (bit-exact; pow within its bar)."""
    rng = np.random.default_rng(77)
    n = 9 << 20                                           # 36 MiB of f32 per array
    a = rng.uniform(0.5, 2.0, n).astype(np.float32); b = rng.uniform(0.5, 2.0, n).astype(np.float32)
    da, db = smhip.to_device(a), smhip.to_device(b)
    tail = 3                                              # vector body + scalar tail
    sub = lambda d, k: sma.DeviceArray(smhip, d.base_ptr, np.float32, (k,), (1,), 0, d._owner)
    # contiguous (2R+1W = 96 MiB), array-scalar (64 MiB + tail: one element short of... keep it inside the window)
    got = smhip.contiguous(sma.OP_ADD, da, db).numpy()
    assert np.array_equal(got, a + b)
    got = smhip.array_scalar(sma.OP_MUL, da, np.float32(1.25)).numpy()
    assert np.array_equal(got, a * np.float32(1.25))
    got = smhip.contiguous(sma.OP_SUB, sub(da, n - tail), sub(db, n - tail)).numpy()
    assert np.array_equal(got, (a - b)[: n - tail])
    # heavy tile kernels (pow): scalar and array exponents
    got = smhip.array_scalar(sma.OP_POW, da, np.float32(2.5)).numpy()
    want = np.power(a.astype(np.float64), 2.5).astype(np.float32)
    assert orc.ulp_diff_f32(got, want).max() <= POW_ULP
    got = smhip.contiguous(sma.OP_POW, da, db).numpy()
    want = np.power(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    assert orc.ulp_diff_f32(got, want).max() <= POW_ULP
    # row kernel: (2304, 4096) * (1, 4096), 72 MiB
    A = a.reshape(2304, 4096); r = b[:4096].reshape(1, 4096)
    dA = sma.DeviceArray(smhip, da.base_ptr, np.float32, A.shape, (4096, 1), 0, da._owner)
    dr = sma.DeviceArray(smhip, db.base_ptr, np.float32, r.shape, (4096, 1), 0, db._owner)
    assert np.array_equal(smhip.binary(sma.OP_MUL, dA, dr).numpy(), A * r)
    # tile kernel: A.T + B on (2304, 4096) -> output (4096, 2304), 108 MiB
    dAT = sma.DeviceArray(smhip, da.base_ptr, np.float32, (4096, 2304), (1, 4096), 0, da._owner)
    dB = sma.DeviceArray(smhip, db.base_ptr, np.float32, (4096, 2304), (2304, 1), 0, db._owner)
    assert np.array_equal(smhip.binary(sma.OP_ADD, dAT, dB).numpy(), A.T + b.reshape(4096, 2304))
    # LDS kernel: (56, 224, 224, 3) + (1, 224, 1, 3): 32.2 MiB in, 32.2 MiB out (just above the 64 MiB floor)
    m = 56 * 224 * 224 * 3
    X = a[:m].reshape(56, 224, 224, 3); y = b[:224 * 3].reshape(1, 224, 1, 3)
    dX = sma.DeviceArray(smhip, da.base_ptr, np.float32, X.shape, (224 * 224 * 3, 224 * 3, 3, 1), 0, da._owner)
    dy = sma.DeviceArray(smhip, db.base_ptr, np.float32, y.shape, (672, 3, 3, 1), 0, db._owner)
    assert np.array_equal(smhip.binary(sma.OP_ADD, dX, dy).numpy(), X + y)
    # fused (a + b) * c: 3R+1W of 16 MiB = 64 MiB
    k = 1 << 22
    got = smhip.fused(sma.OP_ADD, sma.OP_MUL, sub(da, k), sub(db, k), sub(da, k)).numpy()
    assert np.array_equal(got, (a[:k] + b[:k]) * a[:k])
    # the run-time-compiled flat kernels (user Op, fused expression) carry the same policy word
    op = smhip.register_op("(a + b) * 2")
    assert np.array_equal(smhip.contiguous(op, da, db).numpy(), (a + b) * np.float32(2))
    assert np.array_equal(smhip.array_scalar(op, da, np.float32(0.5)).numpy(), (a + np.float32(0.5)) * np.float32(2))
    assert np.array_equal(smhip.fused_expr("(a0 - a1) * 4", da, db).numpy(), (a - b) * np.float32(4))
    total, diff = smhip.fused_expr_sum("a0 - a1", da, db, store=True)      # reads 72 MiB, writes 36 MiB
    assert np.array_equal(diff.numpy(), a - b)
    assert abs(total - float(np.sum((a - b).astype(np.float64)))) <= 1e-9 * n


@pytest.mark.parametrize("dtn", ["f64", "i32", "i64"])
def test_store_policy_keep_arm_other_types(smhip, dtn):
    """The keep-in-cache (`sc1`) store arm for the other element widths: contiguous, array-scalar, row and tile kernels inside the
    64-256 MiB window, bit for bit against numpy (integer + and * wrap in both)."""
    dt = DT[dtn]
    rng = np.random.default_rng(78)
    rows, cols = 2304, 2048 if np.dtype(dt).itemsize == 8 else 4096      # 36 MiB per array
    n = rows * cols
    if dtn == "f64":
        a = rng.uniform(-2.0, 2.0, n); b = rng.uniform(-2.0, 2.0, n)
    else:
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, n, dtype=dt); b = rng.integers(info.min, info.max, n, dtype=dt)
    da, db = smhip.to_device(a), smhip.to_device(b)
    with np.errstate(over="ignore"):
        assert np.array_equal(smhip.contiguous(sma.OP_ADD, da, db).numpy(), a + b)
        s = dt(3)
        assert np.array_equal(smhip.array_scalar(sma.OP_MUL, da, s).numpy(), a * s)
        A = a.reshape(rows, cols); r = b[:cols].reshape(1, cols)
        dA = sma.DeviceArray(smhip, da.base_ptr, dt, A.shape, (cols, 1), 0, da._owner)
        dr = sma.DeviceArray(smhip, db.base_ptr, dt, r.shape, (cols, 1), 0, db._owner)
        assert np.array_equal(smhip.binary(sma.OP_MUL, dA, dr).numpy(), A * r)
        dAT = sma.DeviceArray(smhip, da.base_ptr, dt, (cols, rows), (1, cols), 0, da._owner)
        dB = sma.DeviceArray(smhip, db.base_ptr, dt, (cols, rows), (rows, 1), 0, db._owner)
        assert np.array_equal(smhip.binary(sma.OP_ADD, dAT, dB).numpy(), A.T + b.reshape(cols, rows))


def test_in_place_against_a_transposed_operand_with_ragged_extents(smhip, oracle):
    """smhip_elementwise(ADD, a, b.T, out = a) on extents the tile kernel's patches do not divide (a hanging
    patch is normally pulled back inside and recomputes its neighbour's elements -- with the output in an operand's place
    that would apply the Op twice).  Every element must be computed exactly once."""
    for n, dt in ((1031, np.float32), (517, np.float64), (1031, np.int32)):
        a = gen.gen(dt, n * n, 71, "uniform").reshape(n, n)
        b = gen.gen(dt, n * n, 72, "uniform").reshape(n, n)
        want = oracle.binary(orc.ADD, a, b.T)
        da, db = smhip.to_device(a), smhip.to_device(b)
        smhip.binary(sma.OP_ADD, da, db.view_like(b.T, b), out=da)
        util.assert_same_bits(da.numpy(), want, f"in place, {n} x {n} {np.dtype(dt).name}")
