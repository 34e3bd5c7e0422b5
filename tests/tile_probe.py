"""The LDS tile kernel's three patch shapes against numpy.  SMHIP_TILE_QB=1024 forces the 64 x 1024 B patch with its
row-major walk at test sizes, =512 the 64 x 512 B patch on its diagonal, =128 the short patch of skinny planes; unset, the
library chooses: planes with a q extent of <= 256 bytes take the short patch, and the last case (one operand > 256 MiB) the
wide one.  Transposed / permuted / offset views, one and two turned operands,
f32 / f64 / i32, + and *, a user-defined Op (hipRTC compiles the same body), bit-exact (one correctly rounded operation
per element).                    python tests/tile_probe.py [big]      -- prints "tile_probe ok <cases>".
"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
from tests.golden import gen

DT = {"f32": np.float32, "f64": np.float64, "i32": np.int32}
smhip = sma.load()
rng = np.random.default_rng(77)
user = smhip.register_op("a * b + (T)3")
cases = 0
for t in range(36):
    dtn = ("f32", "f64", "i32")[t % 3]
    dt = DT[dtn]
    nd = 2 if t % 4 else 3
    # extents: whole patches (multiples of 256), ragged edges and extents that are not a multiple of the vector width
    pick = lambda: int(rng.choice([256, 512, 768, int(rng.integers(130, 1500)), 4 * int(rng.integers(40, 300))]))
    dims = [pick(), pick()] if nd == 2 else [int(rng.integers(2, 6)), pick() // 2 + 64, pick() // 2 + 64]
    a = gen.gen(dt, int(np.prod(dims)), 700 + t, "uniform").reshape(dims)
    b = gen.gen(dt, int(np.prod(dims)), 800 + t, "uniform").reshape(dims)
    turn = lambda x: np.transpose(x, list(range(x.ndim - 2)) + [x.ndim - 1, x.ndim - 2])
    kind = t % 5
    if kind == 0: va, vb = turn(a), b.reshape(turn(a).shape)                  # one turned operand
    elif kind == 1: va, vb = a.reshape(turn(b).shape), turn(b)                # ... the right-hand one
    elif kind == 2: va, vb = turn(a), turn(b)                                 # both turned
    elif kind == 3: va, vb = turn(a)[..., 4:-4, 8:-8], turn(b)[..., 4:-4, 8:-8]   # offset bases, ragged patches
    else: va, vb = np.transpose(a), np.transpose(b).copy()                    # full reversal against a dense operand
    if not vb.flags.c_contiguous and kind == 4: vb = np.ascontiguousarray(vb)
    base_b = b if np.shares_memory(vb, b) else vb
    da, db = smhip.to_device(a), smhip.to_device(base_b)
    for opn, f in (("add", np.add), ("mul", np.multiply), (user, None)):
        op = sma.OPS[opn] if isinstance(opn, str) else opn
        got = smhip.binary(op, da.view_like(va, a), db.view_like(vb, base_b)).numpy()
        want = f(va, vb) if f else va * vb + dt(3)
        if not np.array_equal(got, want):
            print("MISMATCH", t, dtn, opn, dims, kind, int((got != want).sum()))
            sys.exit(1)
        cases += 1
# a turned operand against one that does not move along q (one value per row, one value: `dst = src.T` is such a problem)
for t, (P, Q) in enumerate(((768, 1024), (516, 260), (1000, 772))):
    dtn = ("f32", "f64", "i32")[t % 3]
    dt = DT[dtn]
    a = gen.gen(dt, P * Q, 960 + t, "uniform").reshape(Q, P)
    col = gen.gen(dt, P, 970 + t, "uniform").reshape(P, 1)
    da, dcol = smhip.to_device(a), smhip.to_device(col)
    got = smhip.binary(sma.OPS["sub"], da.view_like(a.T, a), dcol).numpy()
    if not np.array_equal(got, a.T - col):
        print("MISMATCH turned - column", dtn, P, Q); sys.exit(1)
    got = smhip.binary(sma.OPS["sub"], dcol, da.view_like(a.T, a)).numpy()
    if not np.array_equal(got, col - a.T):
        print("MISMATCH column - turned", dtn, P, Q); sys.exit(1)
    dst = smhip.empty((P, Q), dt)
    smhip.assign(dst, da.view_like(a.T, a))
    if not np.array_equal(dst.numpy(), a.T):
        print("MISMATCH dst = src.T", dtn, P, Q); sys.exit(1)
    cases += 3
# skinny planes: q extents of 16 ... 64 elements (the short patch when the library chooses), long p extents
for t, (P, Q) in enumerate(((5000, 32), (3108, 16), (4096, 64), (2052, 8), (1024, 24), (640, 48))):
    dtn = ("f32", "f64", "i32")[t % 3]
    dt = DT[dtn]
    a = gen.gen(dt, P * Q, 900 + t, "uniform").reshape(Q, P)   # stored (Q, P), used transposed
    b = gen.gen(dt, P * Q, 950 + t, "uniform").reshape(P, Q)
    da, db = smhip.to_device(a), smhip.to_device(b)
    for opn, f in (("add", np.add), ("mul", np.multiply)):
        got = smhip.binary(sma.OPS[opn], da.view_like(a.T, a), db).numpy()
        if not np.array_equal(got, f(a.T, b)):
            print("MISMATCH skinny", dtn, opn, P, Q)
            sys.exit(1)
        got = smhip.binary(sma.OPS[opn], db.view_like(b.T, b), da).numpy()   # the other way round: long q, short p
        if not np.array_equal(got, f(b.T, a)):
            print("MISMATCH skinny turned", dtn, opn, P, Q)
            sys.exit(1)
        cases += 2
# pow through the tile kernel: its tables share the LDS with the patch (2 / 5 KiB next to 33 / 66 KiB)
for dt, bar in ((np.float32, 1), (np.float64, 1)):
    for dims in ((512, 768), (300, 260)):
        base = rng.uniform(0.05, 30.0, dims).astype(dt)
        e = rng.uniform(-3.0, 3.0, dims[::-1]).astype(dt)
        db_, de_ = smhip.to_device(base), smhip.to_device(e)
        for va, vb, ba, bb in ((base.T, e, base, e), (base.T, e.T.copy().T, base, None)):
            if bb is None:  # both operands turned: e stored transposed
                et = np.ascontiguousarray(e.T)
                det = smhip.to_device(et)
                got = smhip.binary(sma.OPS["pow"], db_.view_like(base.T, base), det.view_like(et.T, et)).numpy()
            else:
                got = smhip.binary(sma.OPS["pow"], db_.view_like(base.T, base), de_).numpy()
            with np.errstate(all="ignore"):
                exact = np.power(base.T.astype(np.longdouble), e.astype(np.longdouble)).astype(dt)
            it = np.int32 if dt == np.float32 else np.int64
            d = np.abs(got.view(it).astype(np.int64) - exact.view(it).astype(np.int64))
            if d.max() > bar:
                print("MISMATCH pow", np.dtype(dt), dims, int(d.max()))
                sys.exit(1)
            cases += 1
# output rows off the 128-byte lines (inner extents like 1031): with the wide patch (SMHIP_TILE_QB=1024, or unforced past the
# Infinity Cache) these take tile_shift_body -- patch rows cut at the output's LINES, each row at its own shift.  One and two
# turned operands, either side, non-commutative Ops, slices of a third axis, operand views at offset bases, a constant direct
# operand (column), every element type; extents around the patch sizes so that the first / last windows hang over the rows' ends.
for t, dims in enumerate(((300, 1031), (257, 517), (1000, 773), (3, 260, 1029), (512, 2049), (640, 1283), (259, 514), (2, 300, 643), (1030, 1030))):
    dtn = ("f32", "f64", "i32")[t % 3]
    dt = DT[dtn]
    P, Q = dims[-2], dims[-1]
    lead = list(dims[:-2])
    a = gen.gen(dt, int(np.prod(dims)), 1200 + t, "uniform").reshape(lead + [Q, P])   # stored (.., Q, P), used turned
    b = gen.gen(dt, int(np.prod(dims)), 1300 + t, "uniform").reshape(lead + [P, Q])
    wide = gen.gen(dt, int(np.prod(lead + [P, Q + 9])), 1400 + t, "uniform").reshape(lead + [P, Q + 9])
    da, db, dwide = smhip.to_device(a), smhip.to_device(b), smhip.to_device(wide)
    turn = lambda x: np.swapaxes(x, -1, -2)
    at, bview = turn(a), wide[..., 5:5 + Q]   # a direct operand at an offset base with a pitch of its own
    for opn, f in (("sub", np.subtract), ("add", np.add)):
        for (x, dx, y, dy) in ((at, da.view_like(at, a), b, db), (b, db, at, da.view_like(at, a)), (at, da.view_like(at, a), bview, dwide.view_like(bview, wide))):
            got = smhip.binary(sma.OPS[opn], dx, dy).numpy()
            if not np.array_equal(got, f(x, y)):
                bad = np.argwhere(got != f(x, y))
                print("MISMATCH rows off the lines", dtn, opn, dims, len(bad), bad[:4].tolist()); sys.exit(1)
            cases += 1
    # both turned
    bt_store = np.ascontiguousarray(turn(b))
    dbt = smhip.to_device(bt_store)
    got = smhip.binary(sma.OPS["sub"], da.view_like(at, a), dbt.view_like(turn(bt_store), bt_store)).numpy()
    if not np.array_equal(got, at - b):
        print("MISMATCH rows off the lines, both turned", dtn, dims); sys.exit(1)
    # a turned operand against one value per row, and the plain transposed copy
    if not lead:
        col = gen.gen(dt, P, 1500 + t, "uniform").reshape(P, 1)
        dcol = smhip.to_device(col)
        got = smhip.binary(sma.OPS["sub"], dcol, da.view_like(at, a)).numpy()
        if not np.array_equal(got, col - at):
            print("MISMATCH rows off the lines, column - turned", dtn, dims); sys.exit(1)
        dst = smhip.empty((P, Q), dt)
        smhip.assign(dst, da.view_like(at, a))
        if not np.array_equal(dst.numpy(), at):
            print("MISMATCH rows off the lines, dst = src.T", dtn, dims); sys.exit(1)
        cases += 2
    cases += 1
if len(sys.argv) > 1 and sys.argv[1] == "big":
    # 8191 x 8703 f32 = 272 MiB per array, rows off the lines: unforced, the wide patch with shifted rows
    P, Q = 8191, 8703
    a = gen.gen(np.float32, P * Q, 3, "uniform").reshape(Q, P)
    b = gen.gen(np.float32, P * Q, 4, "uniform").reshape(P, Q)
    da, db = smhip.to_device(a), smhip.to_device(b)
    got = smhip.binary(sma.OPS["sub"], da.view_like(a.T, a), db).numpy()
    if not np.array_equal(got, a.T - b):
        print("MISMATCH big, rows off the lines")
        sys.exit(1)
    del a, b, da, db, got
    smhip.pool_trim()
    cases += 1
if len(sys.argv) > 1 and sys.argv[1] == "big":
    # 8704 x 8192 f32 = 272 MiB per array: past the Infinity Cache, so the unforced library takes the wide patch
    P, Q = 8704, 8192
    a = gen.gen(np.float32, P * Q, 1, "uniform").reshape(Q, P)
    b = gen.gen(np.float32, P * Q, 2, "uniform").reshape(P, Q)
    da, db = smhip.to_device(a), smhip.to_device(b)
    got = smhip.binary(sma.OPS["add"], da.view_like(a.T, a), db).numpy()
    if not np.array_equal(got, a.T + b):
        print("MISMATCH big")
        sys.exit(1)
    cases += 1
print("tile_probe ok", cases, "SMHIP_TILE_QB =", os.environ.get("SMHIP_TILE_QB", "(unset)"))
