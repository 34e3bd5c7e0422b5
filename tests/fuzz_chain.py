"""Soak for smhip_chain (csrc/chain.hip): random operator chains -- 1 to 12 stages, + - * / in either operand order -- over
random operand FORMS against the oracle's operator-by-operator evaluation, bit for bit: dense arrays, rows (trailing axes),
columns / per-channel values (leading or middle axes), operands periodic in the output with broadcast axes inside the period
(the reference tests' (1,d1,1,d3)), one-element arrays, scalars, and views the one-pass kernel has no index form for
(transposed, stepped, sliced with a pitch: they cut the chain).  Every element type.
    usage: python tests/fuzz_chain.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplemath_amd as sma
from oracle import oracle as orc

DT = [np.float32, np.float32, np.float64, np.int32, np.int64]
OPS = [sma.OP_ADD, sma.OP_SUB, sma.OP_MUL, sma.OP_DIV]
ORC = {sma.OP_ADD: orc.ADD, sma.OP_SUB: orc.SUB, sma.OP_MUL: orc.MUL, sma.OP_DIV: orc.DIV}


def values(rng, shape, dt):
    if np.dtype(dt).kind == "f":
        x = rng.uniform(0.25, 4.0, size=shape) * rng.choice([-1.0, 1.0], size=shape)
        return x.astype(dt)
    x = rng.integers(1, 50, size=shape) * rng.choice([-1, 1], size=shape)
    return x.astype(dt)


def random_shape(rng):
    nd = int(rng.integers(1, 5))
    while True:
        shape = [int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 64, 100, 128, 257])) for _ in range(nd)]
        n = int(np.prod(shape))
        if 1 <= n <= (1 << 19):
            return tuple(shape)


def operand(rng, shape, dt):
    """-> (host view, (base, view) or array or scalar for the device side, label)"""
    nd = len(shape)
    kind = rng.choice(["dense", "dense", "row", "col", "mid", "periodic", "one", "scalar", "T", "step", "pitch", "lower"])
    if kind == "scalar":
        v = dt(rng.integers(1, 5)) if np.dtype(dt).kind != "f" else dt(rng.choice([0.5, 1.5, 2.0, 3.0, -0.75]))
        return v, v, kind
    if kind == "dense":
        x = values(rng, shape, dt)
        return x, x, kind
    if kind == "one":
        x = values(rng, (1,) * int(rng.integers(1, nd + 1)), dt)
        return x, x, kind
    if kind in ("row", "col", "mid", "periodic"):
        keep = np.zeros(nd, dtype=bool)
        if kind == "row":
            keep[int(rng.integers(0, nd)):] = True
        elif kind == "col":
            keep[:int(rng.integers(1, nd + 1))] = True
        elif kind == "mid":
            lo = int(rng.integers(0, nd)); hi = int(rng.integers(lo, nd))
            keep[lo:hi + 1] = True
        else:
            keep = rng.random(nd) < 0.5
        s = tuple(d if k else 1 for d, k in zip(shape, keep))
        x = values(rng, s, dt)
        return x, x, kind
    if kind == "lower":  # fewer dimensions than the result: right-aligned
        k = int(rng.integers(1, nd + 1))
        x = values(rng, shape[nd - k:], dt)
        return x, x, kind
    if kind == "T" and nd >= 2:
        base = values(rng, shape[::-1], dt)
        return base.T, (base, base.T), kind
    if kind == "step":
        wide = list(shape); wide[-1] = shape[-1] * 2 + 1
        base = values(rng, tuple(wide), dt)
        view = base[..., ::2][..., :shape[-1]]
        return view, (base, view), kind
    if kind == "pitch":
        wide = list(shape); wide[-1] = shape[-1] + int(rng.integers(1, 9))
        base = values(rng, tuple(wide), dt)
        off = int(rng.integers(0, wide[-1] - shape[-1] + 1))
        view = base[..., off:off + shape[-1]]
        return view, (base, view), kind
    x = values(rng, shape, dt)
    return x, x, "dense"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    lib = sma.load()
    o = orc.Oracle()
    fused = 0
    trace = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
    for case in range(cases):
        dt = DT[int(rng.integers(0, len(DT)))]
        shape = random_shape(rng)
        n_stages = int(rng.integers(1, 13))
        head_h, head_d, label = operand(rng, shape, dt)
        while np.isscalar(head_h) or not isinstance(head_h, np.ndarray):
            head_h, head_d, label = operand(rng, shape, dt)
        labels = [label]
        stages_h, stages_d = [], []
        keep = []

        def to_dev(x):
            if isinstance(x, tuple):
                base, view = x
                d = lib.to_device(base)
                keep.append(d)
                return d.view_like(view, base)
            if isinstance(x, np.ndarray):
                d = lib.to_device(x)
                keep.append(d)
                return d
            return x

        dhead = to_dev(head_d)
        for _ in range(n_stages):
            h, d, label = operand(rng, shape, dt)
            op = int(rng.choice(OPS))
            swapped = bool(rng.random() < 0.3) and isinstance(h, np.ndarray)
            stages_h.append((op, h, swapped))
            stages_d.append((op, to_dev(d), swapped))
            labels.append(("~" if swapped else "") + "+-*/"[op] + label)
        # oracle: one operator at a time
        r = np.ascontiguousarray(head_h)
        for op, h, swapped in stages_h:
            if isinstance(h, np.ndarray):
                r = o.binary(ORC[op], h, r) if swapped else o.binary(ORC[op], r, h)
            else:
                r = o.array_scalar(ORC[op], np.ascontiguousarray(r).reshape(-1), h).reshape(r.shape)
        if trace:  # which case was in flight when something went wrong on the GPU
            def sh(x):
                return (tuple(x.shape), tuple(x.strides), x.offset) if isinstance(x, sma.DeviceArray) else x
            trace.write(f"case {case}: {np.dtype(dt).name} out {shape} head {sh(dhead)} " + " ".join(f"[{'~' if sw else ''}{'+-*/'[op]} {sh(d)}]" for op, d, sw in stages_d) + "\n")
            trace.flush()
            os.fsync(trace.fileno())
        try:
            got = lib.chain(dhead, *stages_d)
            if trace:
                lib.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"case {case}: {np.dtype(dt).name} {shape} {' '.join(labels)}: {e}")
            raise
        g = got.numpy()
        want = np.ascontiguousarray(r)
        ok = g.shape == want.shape and g.tobytes() == want.tobytes()
        if not ok and np.dtype(dt).kind == "f":  # NaN payloads are not part of the bar
            ok = g.shape == want.shape and np.array_equal(g, want, equal_nan=True)
        if not ok:
            bad = np.flatnonzero(g.reshape(-1) != want.reshape(-1))
            print(f"MISMATCH case {case} seed {seed}: {np.dtype(dt).name} {shape} {' '.join(labels)}: {bad.size} of {g.size} elements, first at {bad[:5]}: "
                  f"{g.reshape(-1)[bad[:3]]} want {want.reshape(-1)[bad[:3]]}")
            return 1
        fused += 1
    print(f"ok: {fused} chains, seed {seed}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
