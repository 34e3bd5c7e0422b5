"""Operator chains in one pass (smhip_chain, csrc/chain.hip) against the ORACLE'S chain: the reference evaluates
`(A * row + B) * 0.5f` as one operator call and one temporary per step (SMArray.h:217-305), so the oracle's answer is its
`binary` / `array_scalar` applied step by step.  Bar: bit-identical (each stage is the single rounded / wrapping operation).
"""
import numpy as np
import pytest

import simplemath_amd as sma
from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64, np.int32, np.int64]
ORC_OP = {sma.OP_ADD: orc.ADD, sma.OP_SUB: orc.SUB, sma.OP_MUL: orc.MUL, sma.OP_DIV: orc.DIV}


def _rand(rng, shape, dt):
    if np.dtype(dt).kind == "f":
        x = rng.uniform(-2.0, 2.0, size=shape).astype(dt)
        x[np.abs(x) < 0.05] = 0.5  # divisors away from zero keep the chain finite; IEEE specials have a test of their own
        return x
    x = rng.integers(-1000, 1000, size=shape).astype(dt)
    x[x == 0] = 7
    return x


def _oracle_chain(o, first, stages):
    r = np.ascontiguousarray(first)
    for st in stages:
        op, x = ORC_OP[st[0]], st[1]
        swapped = len(st) > 2 and st[2]
        if isinstance(x, np.ndarray):
            r = o.binary(op, x, r) if swapped else o.binary(op, r, x)
        else:
            assert not swapped
            r = o.array_scalar(op, r.reshape(-1), x).reshape(r.shape)
    return r


def _device(smhip, x, keep):
    """host operand -> device operand; `x` may be a (base, view) pair for a strided view"""
    if isinstance(x, tuple):
        base, view = x
        d = smhip.to_device(base)
        keep.append(d)
        return d.view_like(view, base), view
    if isinstance(x, np.ndarray):
        d = smhip.to_device(x)
        keep.append(d)
        return d, x
    return x, x


def _run(smhip, oracle, first, stages, what):
    keep = []
    dfirst, hfirst = _device(smhip, first, keep)
    dst, hst = [], []
    for st in stages:
        dx, hx = _device(smhip, st[1], keep)
        dst.append((st[0], dx) + tuple(st[2:]))
        hst.append((st[0], hx) + tuple(st[2:]))
    got = smhip.chain(dfirst, *dst)
    want = _oracle_chain(oracle, hfirst, hst)
    assert list(got.shape) == list(want.shape), what
    util.assert_same_bits(got.numpy(), want, what)


@pytest.mark.parametrize("dt", DTYPES, ids=lambda d: np.dtype(d).name)
def test_chain_row_column_scalar(smhip, oracle, dt):
    """(dense o row) o dense o scalar -- the harness's chain_check -- and its column / scalar / all-dense relatives."""
    rng = np.random.default_rng(11)
    for rows, cols in [(64, 128), (37, 52), (5, 3), (1, 7), (129, 1024)]:
        A, B = _rand(rng, (rows, cols), dt), _rand(rng, (rows, cols), dt)
        row, col = _rand(rng, (1, cols), dt), _rand(rng, (rows, 1), dt)
        one = _rand(rng, (1, 1), dt)
        s = dt(3) if np.dtype(dt).kind != "f" else dt(0.5)
        tag = f"{np.dtype(dt).name} {rows}x{cols}"
        _run(smhip, oracle, A, [(sma.OP_MUL, row), (sma.OP_ADD, B), (sma.OP_MUL, s)], tag + " (A*row+B)*s")
        _run(smhip, oracle, A, [(sma.OP_SUB, col), (sma.OP_DIV, B), (sma.OP_ADD, s)], tag + " (A-col)/B+s")
        _run(smhip, oracle, A, [(sma.OP_ADD, B), (sma.OP_MUL, A), (sma.OP_SUB, B), (sma.OP_DIV, s)], tag + " dense x4")
        _run(smhip, oracle, A, [(sma.OP_DIV, row, True), (sma.OP_SUB, col, True)], tag + " col-(row/A)")
        _run(smhip, oracle, A, [(sma.OP_MUL, one), (sma.OP_ADD, row), (sma.OP_SUB, col)], tag + " one-element operand")
        _run(smhip, oracle, row, [(sma.OP_ADD, col), (sma.OP_MUL, A)], tag + " (row+col)*A: the head broadcasts too")
        _run(smhip, oracle, col, [(sma.OP_MUL, s), (sma.OP_SUB, row, True)], tag + " row-(col*s): no dense operand at all")


@pytest.mark.parametrize("dt", DTYPES, ids=lambda d: np.dtype(d).name)
def test_chain_per_row_and_per_channel_values(smhip, oracle, dt):
    """(A - mean) / std with one value per row, and per channel of an NCHW batch: rows / planes that are a whole number of
    waves' worth of vectors take the wave-uniform index (one scalar load per wave), the others the per-lane one -- both against
    the oracle, with a plane count that makes the channel index wrap."""
    rng = np.random.default_rng(12)
    for shape, small in [((96, 512), (96, 1)), ((33, 1280), (33, 1)), ((40, 260), (40, 1)),
                         ((5, 7, 16, 16), (1, 7, 1, 1)), ((3, 4, 32, 24), (1, 4, 1, 1)), ((2, 3, 10, 26), (1, 3, 1, 1)), ((6, 2, 64, 8), (6, 2, 1, 1))]:
        A = _rand(rng, shape, dt)
        mean, std = _rand(rng, small, dt), _rand(rng, small, dt)
        std[std == 0] = 1
        tag = f"{np.dtype(dt).name} {shape} {small}"
        _run(smhip, oracle, A, [(sma.OP_SUB, mean), (sma.OP_DIV, std)], tag + " (A-mean)/std")
        _run(smhip, oracle, A, [(sma.OP_SUB, mean, True), (sma.OP_MUL, std), (sma.OP_ADD, A)], tag + " (mean-A)*std+A")


@pytest.mark.parametrize("dt", DTYPES, ids=lambda d: np.dtype(d).name)
def test_chain_periodic_4d(smhip, oracle, dt):
    """The reference tests' broadcast pattern, (N,224,224,3) o (1,224,1,3) (tests/add.cpp:59-92), inside a chain; a
    per-channel bias (1,C,1,1); a 3-element period (rows of 3 are not whole vectors)."""
    rng = np.random.default_rng(12)
    big = _rand(rng, (2, 24, 20, 3), dt)
    small = _rand(rng, (1, 24, 1, 3), dt)
    rgb = _rand(rng, (1, 1, 1, 3), dt)
    s = dt(2)
    _run(smhip, oracle, big, [(sma.OP_ADD, small), (sma.OP_MUL, rgb), (sma.OP_SUB, s)], "periodic (1,24,1,3) then rgb")
    x = _rand(rng, (3, 8, 6, 10), dt)
    bias = _rand(rng, (1, 8, 1, 1), dt)
    scale = _rand(rng, (1, 8, 1, 1), dt)
    _run(smhip, oracle, x, [(sma.OP_SUB, bias), (sma.OP_DIV, scale), (sma.OP_MUL, x)], "NCHW (x-mean)/std*x")
    y = _rand(rng, (3, 8, 5, 7), dt)  # 35 elements per channel plane: the splat index is per element
    bias2 = _rand(rng, (1, 8, 1, 1), dt)
    _run(smhip, oracle, y, [(sma.OP_ADD, bias2), (sma.OP_MUL, s)], "NCHW with odd planes")
    lead = _rand(rng, (3, 1, 1, 1), dt)
    _run(smhip, oracle, y, [(sma.OP_MUL, lead), (sma.OP_ADD, bias2)], "per-sample value then per-channel value")


@pytest.mark.parametrize("dt", [np.float32, np.int32], ids=lambda d: np.dtype(d).name)
def test_chain_cut_by_views(smhip, oracle, dt):
    """Transposed and stepped views are not leaves of the one-pass kernel: the chain is cut there and continues."""
    rng = np.random.default_rng(13)
    n = 96
    A, B = _rand(rng, (n, n), dt), _rand(rng, (n, n), dt)
    row = _rand(rng, (1, n), dt)
    wide = _rand(rng, (n, 2 * n), dt)
    s = dt(2)
    _run(smhip, oracle, A, [(sma.OP_ADD, (B, B.T)), (sma.OP_MUL, row), (sma.OP_SUB, s)], "A + B.T first")
    _run(smhip, oracle, A, [(sma.OP_MUL, row), (sma.OP_ADD, (B, B.T)), (sma.OP_SUB, s)], "B.T in the middle")
    _run(smhip, oracle, A, [(sma.OP_MUL, row), (sma.OP_ADD, B), (sma.OP_SUB, (wide, wide[:, ::2]), True)], "stepped view last, swapped")
    _run(smhip, oracle, (B, B.T), [(sma.OP_MUL, row), (sma.OP_ADD, A)], "the head is a transposed view")
    _run(smhip, oracle, (B, B.T), [(sma.OP_MUL, s), (sma.OP_ADD, A)], "transposed head against a scalar")
    _run(smhip, oracle, (wide, wide[:, 3:3 + n]), [(sma.OP_MUL, row), (sma.OP_ADD, A)], "a column block of a wider array as head")


def test_chain_two_rows_or_two_columns_at_the_start(smhip, oracle):
    """Found by tests/fuzz_chain.py (seed 3, case 37: a GPU memory access fault): a chain whose head and first operand are two
    DIFFERENT rows (or columns) -- more small operands of one kind than a kernel variant takes, with nothing before them to
    flush.  That operator runs alone; the chain continues."""
    rng = np.random.default_rng(16)
    for dt in (np.float32, np.int64):
        A = _rand(rng, (257, 32), dt)
        r1, r2, r3 = (_rand(rng, (1, 32), dt) for _ in range(3))
        c1, c2 = _rand(rng, (257, 1), dt), _rand(rng, (257, 1), dt)
        _run(smhip, oracle, r1, [(sma.OP_ADD, r2), (sma.OP_MUL, A), (sma.OP_SUB, r3)], "row + row first")
        _run(smhip, oracle, c1, [(sma.OP_SUB, c2), (sma.OP_MUL, A), (sma.OP_ADD, c1, True)], "column - column first")
        _run(smhip, oracle, r1, [(sma.OP_DIV, r2, True), (sma.OP_ADD, r3), (sma.OP_MUL, c1), (sma.OP_SUB, c2)], "rows then columns, no dense operand")
        _run(smhip, oracle, A, [(sma.OP_MUL, r1), (sma.OP_ADD, r2), (sma.OP_SUB, c1), (sma.OP_DIV, c2), (sma.OP_ADD, r3)], "alternating small operands")


def test_chain_long_and_repeated_operands(smhip, oracle):
    """More stages and more distinct operands than one kernel variant takes (4 dense, 1 row, 1 column): cut and continued."""
    rng = np.random.default_rng(14)
    dt = np.float32
    shape = (48, 64)
    arrs = [_rand(rng, shape, dt) for _ in range(7)]
    rows = [_rand(rng, (1, 64), dt) for _ in range(3)]
    cols = [_rand(rng, (48, 1), dt) for _ in range(2)]
    stages = []
    ops = [sma.OP_ADD, sma.OP_MUL, sma.OP_SUB, sma.OP_DIV]
    for k, x in enumerate(arrs[1:] + rows + cols + [dt(1.5), arrs[0], arrs[0]]):
        stages.append((ops[k % 4], x) + ((True,) if k % 3 == 2 and isinstance(x, np.ndarray) else ()))
    _run(smhip, oracle, arrs[0], stages, "14-stage chain")
    _run(smhip, oracle, arrs[0], [(sma.OP_MUL, arrs[0]), (sma.OP_ADD, arrs[0]), (sma.OP_MUL, rows[0]), (sma.OP_ADD, rows[0])], "a*a+a, row twice")


def test_chain_ieee_specials_and_int_division(smhip, oracle):
    spec = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, 1.0, -1.0, 2.5, 1e-38], dtype=np.float32)
    A = np.tile(spec, (12, 1))
    B = np.ascontiguousarray(A.T)
    row = spec.reshape(1, -1)
    _run(smhip, oracle, A, [(sma.OP_MUL, row), (sma.OP_ADD, B), (sma.OP_DIV, A), (sma.OP_SUB, np.float32(0.5))], "specials")
    ia = np.array([[7, -7, 0, 2147483647, -2147483648, 5]], dtype=np.int32).repeat(6, 0)
    ib = np.array([[2], [-2], [0], [-1], [1], [3]], dtype=np.int32)
    keep = []
    got = smhip.chain(smhip.to_device(ia), (sma.OP_DIV, smhip.to_device(ib)), (sma.OP_MUL, np.int32(3))).numpy()
    # defined where the reference traps (ops.hip.h): x / 0 = 0, INT_MIN / -1 = INT_MIN; everything else C's truncation
    one = smhip.binary(sma.OP_DIV, smhip.to_device(ia), smhip.to_device(ib))
    want = smhip.array_scalar(sma.OP_MUL, one, np.int32(3)).numpy()
    assert np.array_equal(got, want)
    safe = (ib != 0) & ~((ia == -2147483648) & (ib == -1))
    ref = (np.trunc(ia.astype(np.float64) / np.where(ib == 0, 1, ib)).astype(np.int64) * 3).astype(np.int32)
    assert np.array_equal(got[safe], ref[safe])


def test_chain_pieces_and_tail(smhip, oracle, monkeypatch):
    """1-D chains with n % W != 0 (the tail lane) and sizes around the vector / workgroup boundaries."""
    rng = np.random.default_rng(15)
    for dt in (np.float32, np.float64):
        for n in (1, 2, 3, 5, 255, 256, 257, 1023, 4099, 70001):
            a, b = _rand(rng, (n,), dt), _rand(rng, (n,), dt)
            one = _rand(rng, (1,), dt)
            _run(smhip, oracle, a, [(sma.OP_ADD, b), (sma.OP_MUL, dt(0.25)), (sma.OP_SUB, one)], f"1-D n={n}")


def test_chain_config3_size(smhip, oracle):
    """The harness's chain at BASELINE config 3's size: (A * row + B) * 0.5 on 4096 x 4096, bit-identical to the three
    operator calls on the GPU (which the golden and oracle tests pin), and to the oracle on a slice of rows."""
    rows, cols = 4096, 4096
    A = smhip.uniform_f32(rows * cols, 3, -1.0, 1.0)
    B = smhip.uniform_f32(rows * cols, 9, -1.0, 1.0)
    row = smhip.uniform_f32(cols, 4, -1.0, 1.0)
    A2 = sma.DeviceArray(smhip, A.base_ptr, np.float32, (rows, cols), (cols, 1), 0, A._owner)
    B2 = sma.DeviceArray(smhip, B.base_ptr, np.float32, (rows, cols), (cols, 1), 0, B._owner)
    r2 = sma.DeviceArray(smhip, row.base_ptr, np.float32, (1, cols), (cols, 1), 0, row._owner)
    got = smhip.chain(A2, (sma.OP_MUL, r2), (sma.OP_ADD, B2), (sma.OP_MUL, np.float32(0.5))).numpy()
    t = smhip.binary(sma.OP_MUL, A2, r2)
    t = smhip.binary(sma.OP_ADD, t, B2)
    want = smhip.array_scalar(sma.OP_MUL, t, np.float32(0.5)).numpy()
    util.assert_same_bits(got, want, "chain vs three launches")
    hA = oracle.uniform_f32(rows * cols, 3, -1.0, 1.0).reshape(rows, cols)[1000:1016]
    hB = oracle.uniform_f32(rows * cols, 9, -1.0, 1.0).reshape(rows, cols)[1000:1016]
    hr = oracle.uniform_f32(cols, 4, -1.0, 1.0).reshape(1, cols)
    util.assert_same_bits(got[1000:1016], _oracle_chain(oracle, hA, [(sma.OP_MUL, hr), (sma.OP_ADD, hB), (sma.OP_MUL, np.float32(0.5))]), "slice vs oracle")


def test_chain_pow_stages(smhip, oracle):
    """r ^ scalar as a stage (sm::pow of an expression): ^2 fused into the kernel, other exponents by cutting the chain and running
    pow's own evaluation -- bit-identical to the operators called one after the other on the GPU, and within pow's bar of the
    oracle's chain (libm's pow, PowOp<T>::apply, pow.h:8-10)."""
    rng = np.random.default_rng(21)
    for dt in DTYPES:
        A, B = np.abs(_rand(rng, (70, 96), dt)) + dt(1), _rand(rng, (70, 96), dt)
        row = _rand(rng, (1, 96), dt)
        dA, dB, drow = smhip.to_device(A), smhip.to_device(B), smhip.to_device(row)
        e_other = dt(3) if np.dtype(dt).kind == "i" else dt(2.5)
        for e in (dt(2), e_other, dt(1)):
            got = smhip.chain(dA, (sma.OP_SUB, drow), (sma.OP_POW, e), (sma.OP_ADD, dB), (sma.OP_MUL, dt(2))).numpy()
            t = smhip.binary(sma.OP_SUB, dA, drow)
            t = smhip.array_scalar(sma.OP_POW, t, e)
            t = smhip.binary(sma.OP_ADD, t, dB)
            want = smhip.array_scalar(sma.OP_MUL, t, dt(2)).numpy()
            if np.dtype(dt).kind == "f" and e == e_other:  # negative bases ^ 2.5: NaN in both
                assert np.array_equal(np.isnan(got), np.isnan(want))
                ok = ~np.isnan(want)
                util.assert_same_bits(got[ok], want[ok], f"{np.dtype(dt).name} ^ {e}")
            else:
                util.assert_same_bits(got, want, f"{np.dtype(dt).name} ^ {e}")
        # the power first and last in the chain, and against the oracle (squares are exact products)
        got = smhip.chain(dA, (sma.OP_POW, dt(2)), (sma.OP_SUB, dB)).numpy()
        util.assert_same_bits(got, oracle.binary(orc.SUB, oracle.binary(orc.MUL, A, A), B), f"{np.dtype(dt).name} A^2 - B")
        got = smhip.chain(dA, (sma.OP_ADD, dB), (sma.OP_POW, dt(2))).numpy()
        s1 = oracle.binary(orc.ADD, A, B)
        util.assert_same_bits(got, oracle.binary(orc.MUL, s1, s1), f"{np.dtype(dt).name} (A + B)^2")
        got = smhip.chain(drow, (sma.OP_POW, dt(2)), (sma.OP_ADD, dA)).numpy()  # a broadcast head: squared after being written out
        util.assert_same_bits(got, oracle.binary(orc.ADD, oracle.binary(orc.MUL, row, row), A), f"{np.dtype(dt).name} row^2 + A")
    with pytest.raises(RuntimeError):
        smhip.chain(dA, (sma.OP_POW, dB))  # an array exponent is not a chain stage


def test_chain_sum(smhip, oracle):
    """The sum of a chain's value without writing it (smhip_chain_sum): one pass for dense / scalar operands, the chain into a
    temporary and its sum otherwise -- against the oracle's sum of the oracle's chain (fp64 accumulation, any order; integers
    exactly, modulo 2^64), for sizes with and without a tail, the squared error among them."""
    rng = np.random.default_rng(31)
    for dt in DTYPES:
        for shape in [(70, 96), (1, 5), (33, 1), (257, 1031), (1 << 20,)]:
            A, B = _rand(rng, shape, dt), _rand(rng, shape, dt)
            dA, dB = smhip.to_device(A), smhip.to_device(B)
            two = dt(2)
            cases = [[(sma.OP_SUB, dB), (sma.OP_POW, two)],                          # the squared error: one pass
                     [(sma.OP_MUL, dB), (sma.OP_ADD, dA), (sma.OP_MUL, dt(3))],       # dense + scalar stages
                     [(sma.OP_ADD, dt(1)), (sma.OP_SUB, dB, True)]]                   # swapped
            if len(shape) == 2 and shape[1] > 1:
                row = _rand(rng, (1, shape[1]), dt)
                cases.append([(sma.OP_MUL, smhip.to_device(row)), (sma.OP_ADD, dB)])  # a row operand: chain, then sum
            for stages in cases:
                got = smhip.chain_sum(dA, *stages)
                r = np.ascontiguousarray(A)
                for st in stages:
                    if st[0] == sma.OP_POW:  # ^2: one product
                        r = oracle.binary(orc.MUL, r, r)
                    else:
                        x = st[1].numpy() if isinstance(st[1], sma.DeviceArray) else st[1]
                        r = _oracle_chain(oracle, r, [(st[0], x) + tuple(st[2:])])
                want = oracle.sum(np.ascontiguousarray(r).reshape(-1))
                scale = float(np.abs(r.astype(np.float64)).sum())
                assert abs(got - want) <= 1e-15 * scale + 1e-300, (np.dtype(dt).name, shape, got, want)


def test_chain_sum_in_pieces():
    """Very large arrays go out in pieces (internal.h: piece_for); SMHIP_PIECE_LOG2VEC=12 cuts a 300 001-element sum into 19 launches
    whose partials must line up."""
    import os, subprocess, sys
    code = ("import numpy as np, simplemath_amd as sma\n"
            "lib = sma.load()\n"
            "rng = np.random.default_rng(3)\n"
            "for dt in (np.float32, np.float64, np.int32):\n"
            "    a = (rng.uniform(-2, 2, 300001) * (50 if dt == np.int32 else 1)).astype(dt); b = (rng.uniform(-2, 2, 300001) * (50 if dt == np.int32 else 1)).astype(dt)\n"
            "    got = lib.chain_sum(lib.to_device(a), (sma.OP_SUB, lib.to_device(b)), (sma.OP_POW, dt(2)))\n"
            "    d = (a - b); want = float(np.sum((d * d).astype(np.float64)))\n"
            "    assert abs(got - want) <= 1e-12 * abs(want) + 1e-9, (dt, got, want)\n"
            "print('pieces ok')\n")
    env = dict(os.environ, SMHIP_PIECE_LOG2VEC="12", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "pieces ok" in r.stdout, r.stdout + r.stderr


def test_chain_errors(smhip):
    a = smhip.to_device(np.ones((4, 4), np.float32))
    b = smhip.to_device(np.ones((3, 4), np.float32))
    with pytest.raises(RuntimeError):
        smhip.chain(a, (sma.OP_ADD, b))
    with pytest.raises(sma.SmhipError):
        smhip.chain(a, (sma.OP_POW, a))  # pow is not a chain stage


@pytest.mark.parametrize("dt", DTYPES, ids=lambda d: np.dtype(d).name)
def test_fused_expr_with_broadcast_operands(smhip, oracle, dt):
    """smhip_fused_expr_bcast: the run-time compiled n-ary expression over operands that broadcast against each other --
    a row, a column, the reference tests' periodic (1,H,1,C), a one-element array, a transposed view (copied dense first) --
    bit-identical to the oracle's operator-by-operator evaluation (no contraction: each operation rounds alone)."""
    rng = np.random.default_rng(21)
    rows, cols = 45, 72
    A, B = _rand(rng, (rows, cols), dt), _rand(rng, (rows, cols), dt)
    row, col, one = _rand(rng, (1, cols), dt), _rand(rng, (rows, 1), dt), _rand(rng, (1, 1), dt)
    s = dt(3)
    dA, dB, drow, dcol, done = (smhip.to_device(x) for x in (A, B, row, col, one))
    got = smhip.fused_expr_bcast("(a0 * a1 + a2) * s0", dA, drow, dB, scalars=[s]).numpy()
    util.assert_same_bits(got, _oracle_chain(oracle, A, [(sma.OP_MUL, row), (sma.OP_ADD, B), (sma.OP_MUL, s)]), "(A*row+B)*s")
    got = smhip.fused_expr_bcast("(a0 + a1) * (a2 - a3) / a4", dA, dcol, dB, drow, done).numpy()
    want = oracle.binary(orc.DIV, oracle.binary(orc.MUL, oracle.binary(orc.ADD, A, col), oracle.binary(orc.SUB, B, row)), one)
    util.assert_same_bits(got, want, "tree expression")
    # no full-size operand at all: (row, col) -> (rows, cols)
    got = smhip.fused_expr_bcast("a0 - a1 * a2", drow, dcol, done).numpy()
    util.assert_same_bits(got, oracle.binary(orc.SUB, row, oracle.binary(orc.MUL, col, one)), "row - col*one")
    # a transposed view among the operands
    S = _rand(rng, (cols, cols), dt)
    dS = smhip.to_device(S)
    got = smhip.fused_expr_bcast("a0 + a1 * a2", dS, dS.view_like(S.T, S), drow).numpy()
    util.assert_same_bits(got, oracle.binary(orc.ADD, S, oracle.binary(orc.MUL, S.T, row)), "S + S.T*row")
    # 4-D: the reference tests' pattern and a 3-element period, ragged total (n % W != 0 for f32)
    big, small, rgb = _rand(rng, (3, 7, 5, 3), dt), _rand(rng, (1, 7, 1, 3), dt), _rand(rng, (1, 1, 1, 3), dt)
    dbig, dsmall, drgb = smhip.to_device(big), smhip.to_device(small), smhip.to_device(rgb)
    got = smhip.fused_expr_bcast("(a0 + a1) * a2", dbig, dsmall, drgb).numpy()
    util.assert_same_bits(got, oracle.binary(orc.MUL, oracle.binary(orc.ADD, big, small), rgb), "4-D periodic")
    # equal dense shapes fall through to the flat kernel
    got = smhip.fused_expr_bcast("a0 * a1 - a0", dA, dB).numpy()
    util.assert_same_bits(got, oracle.binary(orc.SUB, oracle.binary(orc.MUL, A, B), A), "dense")


def test_fuzz_chain_smoke(smhip):
    """A short run of tests/fuzz_chain.py: random chains of 1-12 stages over random operand forms (dense, row, column,
    periodic, one element, scalar, transposed / stepped / pitched views) and element types, bit for bit against the oracle's
    operator-by-operator evaluation."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_chain.py"), "150", "17"], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
