"""Randomised planes with one small extent (records of k elements <-> k long rows, optionally batched, optionally inside wider
rows) through smhip_elementwise and smhip_copy_strided: the record kernel, the tile kernel's short patch and whatever the
planner falls back to, against numpy (one correctly rounded operation per element).
usage: python tests/fuzz_records.py [cases] [seed]      -- prints the first mismatch and exits 1, else "ok"."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import simplemath_amd as sma
from tests.golden import gen

DT = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64}
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
smhip = sma.load()
OPS = {"add": np.add, "sub": np.subtract, "mul": np.multiply}
for c in range(cases):
    dtn = ("f32", "f64", "i32", "i64")[int(rng.integers(0, 4))]
    dt = DT[dtn]
    k = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 15, 16, 17, 20, 24, 31, 32, 33, 40, 48, 64, 72, 96, 100, 127, 128]))
    n = int(rng.choice([4096, 4097, 5000, 8192, 12345, 40000, 65536, 100003]))
    B = int(rng.choice([1, 1, 1, 2, 5]))
    pad = int(rng.choice([0, 0, 0, 1, 7, 64]))
    opn = ("add", "sub", "mul")[int(rng.integers(0, 3))]
    f = OPS[opn]
    recs = gen.gen(dt, B * n * k, 1000 + c, "uniform").reshape(B, n, k)            # B x (n records of k)
    wide = gen.gen(dt, B * k * (n + pad), 2000 + c, "uniform").reshape(B, k, n + pad)
    rows = wide[:, :, pad // 2: pad // 2 + n]                                      # B x (k rows of n), maybe inside wider rows
    drecs, dwide = smhip.to_device(recs), smhip.to_device(wide)
    rt, wt = np.transpose(recs, (0, 2, 1)), np.transpose(rows, (0, 2, 1))
    form = int(rng.integers(0, 6))
    what = (c, dtn, k, n, B, pad, opn, form)
    if form == 0:    # AoS -> SoA
        got = smhip.binary(sma.OPS[opn], drecs.view_like(rt, recs), dwide.view_like(rows, wide)).numpy(); want = f(rt, rows)
    elif form == 1:  # ... operands exchanged
        got = smhip.binary(sma.OPS[opn], dwide.view_like(rows, wide), drecs.view_like(rt, recs)).numpy(); want = f(rows, rt)
    elif form == 2:  # SoA -> AoS
        got = smhip.binary(sma.OPS[opn], dwide.view_like(wt, wide), drecs).numpy(); want = f(wt, recs)
    elif form == 3:
        got = smhip.binary(sma.OPS[opn], drecs, dwide.view_like(wt, wide)).numpy(); want = f(recs, wt)
    elif form == 4:  # copies
        dst = smhip.empty((B, k, n), dt); smhip.assign(dst, drecs.view_like(rt, recs)); got = dst.numpy(); want = rt
    else:
        dst = smhip.empty((B, n, k), dt); smhip.assign(dst, dwide.view_like(wt, wide)); got = dst.numpy(); want = wt
    if c % 3 == 0:  # ... and a dense batch of small planes read transposed (the planes kernel, or the tile kernel past its limits)
        pn, pm, pB = int(rng.integers(2, 100)), int(rng.integers(2, 100)), int(rng.choice([64, 65, 257, 1000, 3001]))
        x = gen.gen(dt, pB * pn * pm, 3000 + c, "uniform").reshape(pB, pn, pm)
        y = gen.gen(dt, pB * pn * pm, 4000 + c, "uniform").reshape(pB, pm, pn)
        dx, dy = smhip.to_device(x), smhip.to_device(y)
        xt = np.transpose(x, (0, 2, 1))
        g2 = smhip.binary(sma.OPS[opn], dy, dx.view_like(xt, x)).numpy()
        if not np.array_equal(g2, f(y, xt)):
            print("MISMATCH planes", (c, dtn, pn, pm, pB, opn))
            sys.exit(1)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print("MISMATCH", what, "differing:", len(bad), "first at", bad[0])
        sys.exit(1)
print("ok: %d record-plane cases, seed %d" % (cases, seed))
