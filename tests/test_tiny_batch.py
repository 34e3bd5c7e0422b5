"""Operators on tiny arrays are recorded and go out several to a launch (simplemath_amd/csrc/tiny.hip): what a caller can observe
must be exactly what one launch per operator gives -- bit for bit against the oracle -- whatever depends on whatever, whenever
buffers are freed, and the recording must actually batch.  The arrays are those of the reference's smallest benchmarks and tests
(benchmark/add.cpp:4-19, benchmark/pow.cpp:5-28, tests/add.cpp's 2-D / 3-D cases)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import simplemath_amd as sma
from oracle import oracle as orc
from tests import util
from tests.golden import gen

pytestmark = pytest.mark.gpu

DT = [np.float32, np.float64, np.int32, np.int64]
ORC = {sma.OP_ADD: orc.ADD, sma.OP_SUB: orc.SUB, sma.OP_MUL: orc.MUL, sma.OP_DIV: orc.DIV, sma.OP_POW: orc.POW}


@pytest.fixture(scope="module")
def smhip():
    return sma.load()


@pytest.fixture(scope="module")
def oracle():
    return orc.Oracle()


def _vals(dt, n, seed, nonzero=False):
    return gen.gen(dt, n, seed, "nonzero" if nonzero and np.dtype(dt).kind == "i" else "mixed")


def test_independent_operators_share_launches(smhip, oracle):
    """Twenty independent a op b on 5 x 5 arrays: at most two launches carry them, every result right."""
    smhip.synchronize()
    l0, o0 = smhip.tiny_stats()
    keep, want = [], []
    for k in range(20):
        dt = DT[k % 4]
        a, b = _vals(dt, 25, 100 + k).reshape(5, 5), _vals(dt, 25, 200 + k, nonzero=True).reshape(5, 5)
        da, db = smhip.to_device(a), smhip.to_device(b)
        keep.append((da, db))
    smhip.synchronize()  # the uploads of such arrays are recorded too: out with them first
    l1, o1 = smhip.tiny_stats()
    outs = []
    for k, (da, db) in enumerate(keep):
        op = (sma.OP_ADD, sma.OP_SUB, sma.OP_MUL, sma.OP_DIV)[k % 4]
        outs.append((op, smhip.binary(op, da, db)))
    first = outs[0][1].numpy()  # the read-back flushes everything recorded
    l2, o2 = smhip.tiny_stats()
    assert o2 - o1 == 20, (o1, o2)
    assert 1 <= l2 - l1 <= 2, (l1, l2)
    for k, ((op, out), (da, db)) in enumerate(zip(outs, keep)):
        a, b = da.numpy(), db.numpy()
        util.assert_same_bits(out.numpy(), oracle.binary(ORC[op], a, b), f"case {k}")
    assert first.shape == (5, 5)


def test_dependent_operators_keep_their_order(smhip, oracle):
    """c = a + b; d = c * c; e = d - a; f = e / b: each reads what the previous one wrote (a recorded operator is launched before
    one that depends on it is recorded), and an operand overwritten by upload after an operator was recorded keeps its old values
    for that operator."""
    for dt in DT:
        a, b = _vals(dt, 30, 1).reshape(5, 6), _vals(dt, 30, 2, nonzero=True).reshape(5, 6)
        da, db = smhip.to_device(a), smhip.to_device(b)
        smhip.synchronize()
        l0, o0 = smhip.tiny_stats()
        c = smhip.binary(sma.OP_ADD, da, db)
        d = smhip.binary(sma.OP_MUL, c, c)
        e = smhip.binary(sma.OP_SUB, d, da)
        f = smhip.binary(sma.OP_DIV, e, db)
        g2 = smhip.binary(sma.OP_ADD, da, da)  # unrelated: a list of its own in the same launch
        smhip.synchronize()
        l1, o1 = smhip.tiny_stats()
        assert (l1 - l0, o1 - o0) == (1, 5), (l0, l1, o0, o1)  # the dependent four ran as ONE workgroup's list, in call order
        util.assert_same_bits(g2.numpy(), oracle.binary(orc.ADD, a, a), "the unrelated one")
        hc = oracle.binary(orc.ADD, a, b)
        hd = oracle.binary(orc.MUL, hc, hc)
        he = oracle.binary(orc.SUB, hd, a)
        hf = oracle.binary(orc.DIV, he, b)
        util.assert_same_bits(f.numpy(), hf, f"{np.dtype(dt).name} chain of dependent tiny operators")
        util.assert_same_bits(d.numpy(), hd, f"{np.dtype(dt).name} intermediate")
        # write-after-read: g = a + b is recorded, then a gets new values
        g = smhip.binary(sma.OP_ADD, da, db)
        a2 = _vals(dt, 30, 3).reshape(5, 6)
        smhip.upload(da.ptr, a2)
        h = smhip.binary(sma.OP_ADD, da, db)
        util.assert_same_bits(g.numpy(), hc, f"{np.dtype(dt).name} recorded before the upload")
        util.assert_same_bits(h.numpy(), oracle.binary(orc.ADD, a2, b), f"{np.dtype(dt).name} recorded after the upload")
        # write-after-write into the same output buffer
        out = smhip.empty((5, 6), dt)
        smhip.binary(sma.OP_ADD, da, db, out=out)
        smhip.binary(sma.OP_MUL, da, db, out=out)
        util.assert_same_bits(out.numpy(), oracle.binary(orc.MUL, a2, b), f"{np.dtype(dt).name} second write wins")


def test_an_operator_that_depends_on_two_lists_joins_them(smhip, oracle):
    """x = a + b and y = a - b are independent (two lists); z = x * y depends on both: the two lists become one (neither depends on
    the other, so either order is call order) and z follows; w = z + x joins as well -- one launch.  Uploads of tiny arrays are
    recorded copies: upload a, upload b, a + b is one launch, not two packets and a launch.  Two LONG lists are not joined: what
    is recorded goes out first."""
    a, b = _vals(np.float64, 12, 21).reshape(3, 4), _vals(np.float64, 12, 22).reshape(3, 4)
    smhip.synchronize()
    l0, o0 = smhip.tiny_stats()
    da, db = smhip.to_device(a), smhip.to_device(b)   # two recorded uploads
    x = smhip.binary(sma.OP_ADD, da, db)
    y = smhip.binary(sma.OP_SUB, da, db)
    z = smhip.binary(sma.OP_MUL, x, y)
    w = smhip.binary(sma.OP_ADD, z, x)
    smhip.synchronize()
    l1, o1 = smhip.tiny_stats()
    assert (l1 - l0, o1 - o0) == (1, 6), (l0, l1, o0, o1)
    hx, hy = oracle.binary(orc.ADD, a, b), oracle.binary(orc.SUB, a, b)
    hz = oracle.binary(orc.MUL, hx, hy)
    util.assert_same_bits(w.numpy(), oracle.binary(orc.ADD, hz, hx), "diamond")
    # two lists of eight dependent operators each, then one operator that needs both: 17 > 12, so two launches
    smhip.synchronize()
    l0, o0 = smhip.tiny_stats()
    p, q, hp, hq = da, db, a, b
    for k in range(8):
        p = smhip.binary(sma.OP_ADD, p, p); hp = oracle.binary(orc.ADD, hp, hp)
        q = smhip.binary(sma.OP_MUL, q, q); hq = oracle.binary(orc.MUL, hq, hq)
    r = smhip.binary(sma.OP_SUB, p, q)
    smhip.synchronize()
    l1, o1 = smhip.tiny_stats()
    assert (l1 - l0, o1 - o0) == (2, 17), (l0, l1, o0, o1)
    util.assert_same_bits(r.numpy(), oracle.binary(orc.SUB, hp, hq), "two long lists")
    # a long dependent sequence: 40 operators in a row, each reading the previous result (lists are cut at 30 operators)
    r, h = da, a
    for k in range(40):
        op = (sma.OP_ADD, sma.OP_MUL, sma.OP_SUB)[k % 3]
        r = smhip.binary(op, r, db)
        h = oracle.binary(ORC[op], h, b)
    util.assert_same_bits(r.numpy(), h, "40 dependent operators")


def test_results_freed_while_recorded(smhip, oracle):
    """The benchmark bodies' pattern: `auto result = a op b` dies at the end of every iteration.  The freed block returns to the
    pool after the launch that writes it -- the next result gets another block, the operators batch, nothing leaks, and a later
    result in a reused block is right."""
    a, b = _vals(np.float32, 25, 7).reshape(5, 5), _vals(np.float32, 25, 8).reshape(5, 5)
    da, db = smhip.to_device(a), smhip.to_device(b)
    smhip.synchronize()
    in_use0, _ = smhip.pool_stats()
    l0, o0 = smhip.tiny_stats()
    for _ in range(100):
        r = smhip.binary(sma.OP_ADD, da, db)
        del r
    last = smhip.binary(sma.OP_MUL, da, db)
    util.assert_same_bits(last.numpy(), oracle.binary(orc.MUL, a, b), "after 100 dropped results")
    l1, o1 = smhip.tiny_stats()
    assert o1 - o0 == 101
    assert l1 - l0 <= 6, (l0, l1)  # 30 to a launch
    del last
    smhip.synchronize()
    in_use1, _ = smhip.pool_stats()
    assert in_use1 == in_use0, (in_use0, in_use1)


def test_every_form_against_the_oracle(smhip, oracle):
    """Views, broadcasts, scalars, host-built operands, integer pow -- each recorded next to unrelated ones."""
    rng = np.random.default_rng(5)
    pending = []
    for t in range(60):
        dt = DT[t % 4]
        kind = t % 6
        if kind == 0:  # broadcast: (3, 4, 5) o (4, 1)
            a, b = _vals(dt, 60, 300 + t).reshape(3, 4, 5), _vals(dt, 4, 400 + t, nonzero=True).reshape(4, 1)
            op = (sma.OP_ADD, sma.OP_DIV)[t % 2]
            got = smhip.binary(op, smhip.to_device(a), smhip.to_device(b))
            want = oracle.binary(ORC[op], a, b)
        elif kind == 1:  # a transposed view against a dense array
            a, b = _vals(dt, 35, 300 + t).reshape(5, 7), _vals(dt, 35, 400 + t).reshape(7, 5)
            da = smhip.to_device(a)
            got = smhip.binary(sma.OP_SUB, da.view_like(a.T, a), smhip.to_device(b))
            want = oracle.binary(orc.SUB, a.T, b)
        elif kind == 2:  # array op scalar
            a = _vals(dt, 17, 300 + t)
            s = dt(3) if np.dtype(dt).kind == "i" else dt(1.5)
            op = (sma.OP_MUL, sma.OP_SUB, sma.OP_DIV)[t % 3]
            got = smhip.array_scalar(op, smhip.to_device(a), s)
            want = oracle.array_scalar(ORC[op], a, s)
        elif kind == 3:  # integer pow (array ^ scalar), floats: a plain product
            a = np.abs(_vals(dt, 10, 300 + t)) % 7 + 1 if np.dtype(dt).kind == "i" else _vals(dt, 10, 300 + t)
            a = a.astype(dt)
            if np.dtype(dt).kind == "i":
                got = smhip.array_scalar(sma.OP_POW, smhip.to_device(a), dt(3))
                want = oracle.array_scalar(orc.POW, a, dt(3))
            else:
                got = smhip.contiguous(sma.OP_MUL, smhip.to_device(a), smhip.to_device(a))
                want = oracle.contiguous(orc.MUL, a, a)
        elif kind == 4:  # a host-built operand riding in the call (smhip_elementwise_inline)
            a, b = _vals(dt, 25, 300 + t).reshape(5, 5), _vals(dt, 25, 400 + t).reshape(5, 5)
            got = smhip.binary_inline(sma.OP_ADD, a, smhip.to_device(b))
            want = oracle.binary(orc.ADD, a, b)
        else:  # a slice with an offset base and a pitch
            a, b = _vals(dt, 80, 300 + t).reshape(8, 10), _vals(dt, 18, 400 + t).reshape(3, 6)
            da = smhip.to_device(a)
            view = a[2:5, 3:9]
            got = smhip.binary(sma.OP_MUL, da.view_like(view, a), smhip.to_device(b))
            want = oracle.binary(orc.MUL, view, b)
        pending.append((t, got, want))
    for t, got, want in pending:
        util.assert_same_bits(got.numpy(), np.ascontiguousarray(want), f"form {t}")


def test_switch_off_and_large_arrays_unaffected(smhip):
    """SMHIP_TINY_BATCH=0 records nothing; arrays above 4096 results are launched as before."""
    code = ("import numpy as np, simplemath_amd as sma\n"
            "lib = sma.load()\n"
            "a = lib.to_device(np.arange(25, dtype=np.float32)); b = lib.to_device(np.ones(25, dtype=np.float32))\n"
            "r = [lib.contiguous(sma.OP_ADD, a, b) for _ in range(10)]\n"
            "assert np.array_equal(r[-1].numpy(), np.arange(25, dtype=np.float32) + 1)\n"
            "print('tiny', lib.tiny_stats())\n")
    env = dict(os.environ, SMHIP_TINY_BATCH="0", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "tiny (0, 0)" in r.stdout, r.stdout + r.stderr
    # a caller that holds the library's stream handle orders its own kernels by what it has seen enqueued: nothing is recorded
    code2 = code.replace("lib = sma.load()\n", "lib = sma.load()\nimport ctypes\nh = ctypes.c_void_p(0)\nassert lib.c.smhip_get_stream(ctypes.byref(h)) == 0 and h.value\n")
    env2 = {k: v for k, v in env.items() if k != "SMHIP_TINY_BATCH"}
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300, env=env2)
    assert r.returncode == 0 and "tiny (0, 0)" in r.stdout, r.stdout + r.stderr
    l0, o0 = smhip.tiny_stats()
    a = smhip.to_device(np.arange(8192, dtype=np.float32))
    r = smhip.contiguous(sma.OP_ADD, a, a)
    assert np.array_equal(r.numpy(), np.arange(8192, dtype=np.float32) * 2)
    assert smhip.tiny_stats() == (l0, o0)
