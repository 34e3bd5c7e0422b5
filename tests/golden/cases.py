"""Case lists shared by make_golden.py (records the reference's outputs) and the
parity tests (replay them through the oracle and the HIP path).

Every case is plain data.  Inputs come from tests/golden/gen.py; a fixture
stores the sha256 of the input bytes next to the expected output so drift in
the generator cannot go unnoticed.
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import gen

DT = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64}
BINOPS = ["add", "sub", "mul", "div"]

# ---------------------------------------------------------------- contiguous
# handle_contiguous_arrays (calculate.h:101-134): n crosses the 8-wide body /
# scalar-tail boundary, the 64-lane wave, the 1024-element chunk.
CONTIG_N = [1, 7, 8, 9, 64, 65, 1023, 1025]


def contiguous_cases():
    out = []
    seed = 1000
    for dt in ("f32", "f64", "i32"):
        for op in BINOPS:
            for n in CONTIG_N:
                seed += 1
                kb = "nonzero" if (dt == "i32" and op == "div") else "mixed"
                out.append({"id": f"contig-{dt}-{op}-{n}", "dtype": dt, "op": op, "n": n,
                            "seed_a": seed, "seed_b": seed + 50000, "kind_a": "mixed", "kind_b": kb})
    return out


def contiguous_inputs(c):
    a = gen.gen(DT[c["dtype"]], c["n"], c["seed_a"], c["kind_a"])
    b = gen.gen(DT[c["dtype"]], c["n"], c["seed_b"], c["kind_b"])
    return a, b


# ----------------------------------------------------------------- broadcast
# element_wise_op's general loop (calculate.h:16-96) behind sm::broadcast
# (SMUtils.h:34-99).  View ops: ["T"] = SMArray::transpose() (reverse axes),
# ["index", k] = operator()(k, SLICE_ALL...) (drop dim 0 at index k),
# ["slice", dim, start, end] = SLICE(start, end) on `dim`.
def _b(id_, dt, op, ashape, bshape, aview=(), bview=(), big=False):
    return {"id": id_, "dtype": dt, "op": op, "a_shape": list(ashape), "b_shape": list(bshape),
            "a_view": [list(v) for v in aview], "b_view": [list(v) for v in bview], "big": big}


def broadcast_cases():
    cs = []
    # config-3 shape in miniature, both orientations, every op / dtype
    for dt in ("f32", "f64", "i32"):
        for op in BINOPS:
            cs.append(_b(f"rowvec-{dt}-{op}", dt, op, (37, 64), (1, 64)))
            cs.append(_b(f"colvec-{dt}-{op}", dt, op, (37, 64), (37, 1)))
    # rank padding: (64,) against (37,64); scalar-like (1,) against 2-D
    cs.append(_b("rankpad-f32-add", "f32", "add", (37, 64), (64,)))
    cs.append(_b("rankpad-rev-f32-sub", "f32", "sub", (64,), (37, 64)))
    cs.append(_b("one-f32-mul", "f32", "mul", (9, 11), (1,)))
    cs.append(_b("one-one-f32-mul", "f32", "mul", (9, 11), (1, 1)))
    # both sides broadcast
    cs.append(_b("outer-f32-add", "f32", "add", (5, 1, 7), (1, 6, 7)))
    cs.append(_b("outer-f32-mul", "f32", "mul", (33, 1), (1, 65)))
    cs.append(_b("outer-i32-mul", "i32", "mul", (33, 1), (1, 65)))
    # the reference's own 4-D view + (1,d1,1,d3) scenario (tests/add.cpp:59-92)
    for op in BINOPS:
        cs.append(_b(f"view4d-f32-{op}", "f32", op, (3, 9, 10, 3), (1, 9, 1, 3), aview=[["index", 1]]))
    # transposed views
    cs.append(_b("T-both-f32-add", "f32", "add", (13, 17), (13, 17), aview=[["T"]], bview=[["T"]]))
    cs.append(_b("T-a-f32-sub", "f32", "sub", (13, 17), (17, 13), aview=[["T"]]))
    cs.append(_b("T-b-f64-mul", "f64", "mul", (17, 13), (13, 17), bview=[["T"]]))
    cs.append(_b("T-3d-i32-add", "i32", "add", (4, 5, 6), (6, 5, 4), bview=[["T"]]))
    cs.append(_b("T-bcast-f32-div", "f32", "div", (8, 1), (12, 8), bview=[["T"]]))
    # sliced views (non-dense outer stride, dense inner)
    cs.append(_b("slice-f32-add", "f32", "add", (20, 32), (20, 32), aview=[["slice", 0, 3, 11]], bview=[["slice", 0, 5, 13]]))
    cs.append(_b("slice-inner-f32-mul", "f32", "mul", (10, 40), (10, 1), aview=[["slice", 1, 4, 36]]))
    cs.append(_b("slice-index-f64-sub", "f64", "sub", (6, 7, 8), (7, 8), aview=[["index", 2]]))
    # 5-D / 6-D (MAX_NDIM, helpers.h:4)
    cs.append(_b("nd5-f32-add", "f32", "add", (2, 3, 1, 5, 4), (1, 3, 6, 1, 4)))
    cs.append(_b("nd6-f32-mul", "f32", "mul", (2, 1, 3, 1, 4, 5), (1, 3, 1, 2, 1, 5)))
    cs.append(_b("nd6-i32-sub", "i32", "sub", (2, 3, 2, 2, 3, 2), (2, 3, 2, 2, 3, 2), bview=[["T"]]))
    # odd inner extents (no 16-byte alignment of rows)
    cs.append(_b("odd-f32-add", "f32", "add", (31, 33), (1, 33)))
    cs.append(_b("odd3-f32-add", "f32", "add", (7, 5, 3), (5, 1)))
    cs.append(_b("odd-i32-div", "i32", "div", (31, 33), (31, 1)))
    # big: crosses CHUNK_SIZE 1024 (macros.h:16) and the n > 100000 OpenMP gate (calculate.h:47)
    cs.append(_b("big-f32-mul", "f32", "mul", (300, 352), (1, 352), big=True))
    cs.append(_b("big-view4d-f32-add", "f32", "add", (2, 224, 224, 3), (1, 224, 1, 3), aview=[["index", 0]], big=True))
    cs.append(_b("big-T-f32-add", "f32", "add", (301, 353), (353, 301), bview=[["T"]], big=True))
    cs.append(_b("big-i32-sub", "i32", "sub", (350, 1, 101), (1, 3, 101), big=True))
    return cs


def apply_view(base: np.ndarray, ops) -> np.ndarray:
    v = base
    for op in ops:
        if op[0] == "T":
            v = v.transpose()
        elif op[0] == "index":
            v = v[op[1]]
        elif op[0] == "slice":
            sl = [slice(None)] * v.ndim
            sl[op[1]] = slice(op[2], op[3])
            v = v[tuple(sl)]
        else:
            raise ValueError(op)
    return v


def _seed_of(case_id: str, salt: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{case_id}:{salt}".encode()).digest()[:4], "little")


def broadcast_inputs(c):
    """-> (a_base, a_view, b_base, b_view): dense bases and the numpy views on them."""
    dt = DT[c["dtype"]]
    na = int(np.prod(c["a_shape"]))
    nb = int(np.prod(c["b_shape"]))
    kind_b = "nonzero" if (c["dtype"] == "i32" and c["op"] == "div") else "mixed"
    a = gen.gen(dt, na, _seed_of(c["id"], 1), "mixed").reshape(c["a_shape"])
    b = gen.gen(dt, nb, _seed_of(c["id"], 2), kind_b).reshape(c["b_shape"])
    return a, apply_view(a, c["a_view"]), b, apply_view(b, c["b_view"])


# -------------------------------------------------------------- array-scalar
SCALAR_N = [1, 8, 9, 1001]
SCALARS = {"f32": [2.0, -0.5, 0.0, float("inf")], "f64": [2.0, -0.5], "i32": [2, -3, 65536]}


def scalar_cases():
    out = []
    seed = 7000
    for dt in ("f32", "f64", "i32"):
        for op in BINOPS:
            for n in SCALAR_N:
                for s in SCALARS[dt]:
                    if dt == "i32" and op == "div" and s in (0, -1):
                        continue  # the reference would trap
                    seed += 1
                    out.append({"id": f"scalar-{dt}-{op}-{n}-{s}", "dtype": dt, "op": op, "n": n,
                                "scalar": s, "seed": seed})
    return out


def scalar_input(c):
    return gen.gen(DT[c["dtype"]], c["n"], c["seed"], "mixed")


# ------------------------------------------------------------------- int pow
IPOW_EXPS = [0, 1, 2, 3, 5, 10, 13, 31, 32, 33, 100, 2147483647, -1, -2, -3, -2147483648]
IPOW_N = [1, 7, 8, 9, 1000, 1003]


def ipow_cases():
    out = []
    seed = 9000
    for n in IPOW_N:
        for e in IPOW_EXPS:
            seed += 1
            out.append({"id": f"ipow-{n}-{e}", "n": n, "exp": e, "seed": seed})
    return out


def ipow_input(c):
    a = gen.gen(np.int32, c["n"], c["seed"], "uniform") % np.int32(60) - np.int32(30)
    k = min(len(gen.I32_SPECIALS), c["n"])
    if c["n"] >= 64:
        a[:k] = gen.I32_SPECIALS[:k]
    return a.astype(np.int32)


# ----------------------------------------------------------------------- dot
DOT_N = [1, 5, 8, 9, 64, 1000, 4099]


def dot_cases():
    out = []
    seed = 11000
    for dt in ("f32", "f64", "i32"):
        for n in DOT_N:
            seed += 1
            out.append({"id": f"dot-{dt}-{n}", "dtype": dt, "n": n, "seed": seed})
    return out


def dot_inputs(c):
    kind = "wide" if c["dtype"] == "i32" else "uniform"
    a = gen.gen(DT[c["dtype"]], c["n"], c["seed"], kind)
    b = gen.gen(DT[c["dtype"]], c["n"], c["seed"] + 77, kind)
    return a, b


# complex and generic-integer dot (product.h:168-224, :8-20) -> tests/golden/dot_extra.npz
CDOT_N = [1, 2, 3, 4, 5, 8, 33, 1000]
GDOT_N = [1, 7, 8, 9, 100, 4099]
GDOT_DTYPES = ["int8", "uint8", "int16", "uint16", "uint32", "uint64"]


def cdot_cases():
    return [{"id": f"cdot-{n}", "n": n, "seed": 16000 + i} for i, n in enumerate(CDOT_N)]


def cdot_inputs(c):
    import numpy as np
    ar, ai = gen.gen(np.float64, c["n"], c["seed"], "uniform"), gen.gen(np.float64, c["n"], c["seed"] + 1, "uniform")
    br, bi = gen.gen(np.float64, c["n"], c["seed"] + 2, "uniform"), gen.gen(np.float64, c["n"], c["seed"] + 3, "uniform")
    return (ar + 1j * ai).astype(np.complex128), (br + 1j * bi).astype(np.complex128)


CDOT32_N = [1, 2, 5, 100, 4099]


def cdot32_cases():
    return [{"id": f"cdot32-{n}", "n": n, "seed": 18000 + i} for i, n in enumerate(CDOT32_N)]


def cdot32_inputs(c):
    import numpy as np
    ar, ai = gen.gen(np.float32, c["n"], c["seed"], "uniform"), gen.gen(np.float32, c["n"], c["seed"] + 1, "uniform")
    br, bi = gen.gen(np.float32, c["n"], c["seed"] + 2, "uniform"), gen.gen(np.float32, c["n"], c["seed"] + 3, "uniform")
    return (ar + 1j * ai).astype(np.complex64), (br + 1j * bi).astype(np.complex64)


def gdot_cases():
    out, seed = [], 17000
    for dt in GDOT_DTYPES:
        for n in GDOT_N:
            seed += 1
            out.append({"id": f"gdot-{dt}-{n}", "dtype": dt, "n": n, "seed": seed})
    return out


def gdot_inputs(c):
    import numpy as np
    dt = np.dtype(c["dtype"])
    rng = np.random.default_rng(c["seed"])
    info = np.iinfo(dt)
    a = rng.integers(info.min, info.max, size=c["n"], dtype=dt, endpoint=True)
    b = rng.integers(info.min, info.max, size=c["n"], dtype=dt, endpoint=True)
    return a, b


# ----------------------------------------------------------------- float pow
POWF_EXPS = [2.5, 2.0, 3.0, 0.5, -1.0, -2.5, 0.0, 1.0, 1.5, 7.0, -3.0, 0.3333333432674408, 10.25, 100.0, -100.0,
             1e-3, float("inf"), float("-inf"), float("nan")]


def powf_cases():
    out = []
    for i, e in enumerate(POWF_EXPS):
        out.append({"id": f"powf-pos-{e}", "exp": e, "n": 1024, "seed": 13000 + i, "kind": "positive"})
        out.append({"id": f"powf-mixed-{e}", "exp": e, "n": 256, "seed": 14000 + i, "kind": "mixed"})
        out.append({"id": f"powf-wide-{e}", "exp": e, "n": 512, "seed": 15000 + i, "kind": "wide"})
    return out


def powf_input(c):
    return gen.gen(np.float32, c["n"], c["seed"], c["kind"])


# ------------------------------------------------- the sizes the reference benchmarks and tests at
# million_check: ones<float>(1'000'000) + ones (benchmark/add.cpp:21-29); the scalar forms at 100 003 (just past the
# n > 100 000 OpenMP gate, calculate.h:152) and 1 000 000; tests/pow.cpp:46-61: empty<int>(1000, 1000, 2) filled with 5, ^3.
# Recorded as 64-value head + tail and the sha256 of the full output (make_golden.py bench_sizes -> bench_sizes.npz).
def bench_size_cases():
    return [
        {"id": "million-ones-f32-add", "kind": "contig", "dtype": "f32", "op": "add", "n": 1_000_000, "fill": "ones"},
        {"id": "million-f32-add", "kind": "contig", "dtype": "f32", "op": "add", "n": 1_000_000, "seed_a": 21001, "seed_b": 21002},
        {"id": "million-f32-div", "kind": "contig", "dtype": "f32", "op": "div", "n": 1_000_000, "seed_a": 21003, "seed_b": 21004},
        {"id": "million-i32-mul", "kind": "contig", "dtype": "i32", "op": "mul", "n": 1_000_000, "seed_a": 21005, "seed_b": 21006},
        {"id": "scalar-f32-mul-100003", "kind": "scalar", "dtype": "f32", "op": "mul", "n": 100_003, "scalar": 2.5, "seed": 21011},
        {"id": "scalar-f32-div-1000000", "kind": "scalar", "dtype": "f32", "op": "div", "n": 1_000_000, "scalar": 3.0, "seed": 21012},
        {"id": "scalar-f64-add-100003", "kind": "scalar", "dtype": "f64", "op": "add", "n": 100_003, "scalar": 0.1, "seed": 21013},
        {"id": "scalar-i32-sub-1000000", "kind": "scalar", "dtype": "i32", "op": "sub", "n": 1_000_000, "scalar": 7, "seed": 21014},
        {"id": "ipow-fives-2000000-3", "kind": "ipow", "n": 2_000_000, "exp": 3, "fill": 5},
        {"id": "ipow-2000000-3", "kind": "ipow", "n": 2_000_000, "exp": 3, "seed": 21021},
        {"id": "ipow-2000000--2", "kind": "ipow", "n": 2_000_000, "exp": -2, "seed": 21022},
    ]


def bench_size_inputs(c):
    """-> (a, b or None)"""
    if c["kind"] == "contig":
        dt = DT[c["dtype"]]
        if c.get("fill") == "ones":
            return np.ones(c["n"], dtype=dt), np.ones(c["n"], dtype=dt)
        kb = "nonzero" if c["dtype"] == "i32" and c["op"] == "div" else "uniform"
        return gen.gen(dt, c["n"], c["seed_a"], "uniform"), gen.gen(dt, c["n"], c["seed_b"], kb)
    if c["kind"] == "scalar":
        return gen.gen(DT[c["dtype"]], c["n"], c["seed"], "uniform"), None
    if "fill" in c:
        return np.full(c["n"], c["fill"], dtype=np.int32), None
    return (gen.gen(np.int32, c["n"], c["seed"], "uniform") % np.int32(60) - np.int32(30)).astype(np.int32), None


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
