#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference.

Run in the build container only (it needs /root/reference, which never travels):

    make -C oracle all ref && python -m tests.golden.make_golden

For every case in tests/golden/cases.py the inputs come from tests/golden/gen.py
and the expected output is whatever the reference's own templates return
through oracle/_ref/libsmref.so (oracle/ref_shim.cpp).  Outputs:

    tests/golden/contiguous.npz   handle_contiguous_arrays   (calculate.h:101-134)
    tests/golden/broadcast.npz    sm::broadcast + element_wise_op (SMUtils.h:34-99, calculate.h:5-99)
    tests/golden/scalar.npz       array_scalar_op            (calculate.h:137-169)
    tests/golden/ipow.npz         PowOp<int> via array_scalar_op (pow.h:70-81, crafted_pow.h:54-103)
    tests/golden/dot.npz          dot_product<T>             (product.h)
    tests/golden/dot_extra.npz    dot_product<std::complex<double>> and the generic dot_product<T> (product.h:168-224, :8-20)
    tests/golden/bench_sizes.npz  the contiguous / scalar / int-pow loops at the sizes the reference benchmarks and tests at
                                  (benchmark/add.cpp:21-29: N = 1 000 000; calculate.h:152's gate: 100 003; tests/pow.cpp:46-61: 2 000 000)
    tests/golden/powf.npz         PowOp<float>::apply = glibc powf (pow.h:8-10) AND the
                                  correctly-rounded value computed in fp64 (the parity target;
                                  the reference itself pins no float pow -- SURVEY 8c)

Each .npz holds, per case id: "<id>/out" (expected output, or for `big` cases a
64-value head + tail sample), "<id>/sha" (sha256 of the full output bytes) and
"<id>/in_sha" (sha256 of the input bytes).  reference_kat.json (the reference's
own 32 gtest cases restated as data) is written by hand, not by this script.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as orc  # noqa: E402
from tests.golden import cases  # noqa: E402


def _sha_arr(x):
    return np.frombuffer(bytes.fromhex(cases.digest(x)), dtype=np.uint8)


def _put(store, cid, out, ins, big=False):
    flat = np.ascontiguousarray(out).reshape(-1)
    store[f"{cid}/sha"] = _sha_arr(flat)
    store[f"{cid}/in_sha"] = np.frombuffer(bytes.fromhex(cases.digest(*ins)), dtype=np.uint8)
    store[f"{cid}/shape"] = np.array(out.shape, dtype=np.int64)
    if big:
        store[f"{cid}/out"] = np.concatenate([flat[:64], flat[-64:]])
    else:
        store[f"{cid}/out"] = flat


def make_dot_extra(ref):
    st = {}
    for c in cases.cdot_cases():  # what the compiled reference returns, n = 1 (scalar tail only) .. 1000 (AVX body: doubled sums)
        a, b = cases.cdot_inputs(c)
        r = ref.dot_c64(a, b)
        _put(st, c["id"], np.array([r.real, r.imag], dtype=np.float64), (a.view(np.float64), b.view(np.float64)))
    for c in cases.cdot32_cases():
        a, b = cases.cdot32_inputs(c)
        r = ref.dot_c32(a, b)
        _put(st, c["id"], np.array([r.real, r.imag], dtype=np.float32), (a.view(np.float32), b.view(np.float32)))
    for c in cases.gdot_cases():
        a, b = cases.gdot_inputs(c)
        _put(st, c["id"], np.array([ref.dot_int(a, b)], dtype=a.dtype), (a, b))
    np.savez_compressed(os.path.join(HERE, "dot_extra.npz"), **st)
    print("dot_extra:", len(st) // 4, "cases")


def make_bench_sizes(ref):
    """The reference's own outputs at the sizes it benchmarks and tests at (cases.bench_size_cases): head + tail + sha256."""
    st = {}
    for c in cases.bench_size_cases():
        a, b = cases.bench_size_inputs(c)
        if c["kind"] == "contig":
            out = ref.elementwise(orc.OPS[c["op"]], a, [1], b, [1], [c["n"]])  # ndim == 1: handle_contiguous_arrays (calculate.h:10)
            _put(st, c["id"], out, (a, b), big=True)
        elif c["kind"] == "scalar":
            _put(st, c["id"], ref.array_scalar(orc.OPS[c["op"]], a, c["scalar"]), (a,), big=True)
        else:
            _put(st, c["id"], ref.array_scalar(orc.POW, a, c["exp"]), (a,), big=True)
    np.savez_compressed(os.path.join(HERE, "bench_sizes.npz"), **st)
    print("bench_sizes:", len(st) // 4, "cases")


def main():
    ref = orc.Reference()
    if sys.argv[1:] == ["dot_extra"]:  # only the file added in round 3 (the others stay byte-identical in git)
        make_dot_extra(ref)
        return
    if sys.argv[1:] == ["bench_sizes"]:  # only the file added in round 4
        make_bench_sizes(ref)
        return
    OPS = orc.OPS

    st = {}
    for c in cases.contiguous_cases():
        a, b = cases.contiguous_inputs(c)
        n = c["n"]
        out = ref.elementwise(OPS[c["op"]], a, [1], b, [1], [n])
        _put(st, c["id"], out, (a, b))
    np.savez_compressed(os.path.join(HERE, "contiguous.npz"), **st)
    print("contiguous:", len(st) // 4, "cases")

    st = {}
    for c in cases.broadcast_cases():
        abase, av, bbase, bv = cases.broadcast_inputs(c)
        res = ref.broadcast(av.shape, orc.elem_strides(av), bv.shape, orc.elem_strides(bv))
        assert res is not None, c["id"]
        shape, sa, sb, total = res
        assert len(shape) > 1, "1-D strided calls are UB in the reference (calculate.h:10)"
        af, aoff = orc._base_and_offset(av)
        bf, boff = orc._base_and_offset(bv)
        out = ref.elementwise(OPS[c["op"]], af[aoff:], sa, bf[boff:], sb, shape).reshape(shape)
        st[f"{c['id']}/strides_a"] = np.array(sa, dtype=np.int64)
        st[f"{c['id']}/strides_b"] = np.array(sb, dtype=np.int64)
        _put(st, c["id"], out, (abase, bbase), big=c["big"])
    np.savez_compressed(os.path.join(HERE, "broadcast.npz"), **st)
    print("broadcast:", len(st) // 6, "cases")

    st = {}
    for c in cases.scalar_cases():
        a = cases.scalar_input(c)
        out = ref.array_scalar(OPS[c["op"]], a, c["scalar"])
        _put(st, c["id"], out, (a,))
    np.savez_compressed(os.path.join(HERE, "scalar.npz"), **st)
    print("scalar:", len(st) // 4, "cases")

    st = {}
    for c in cases.ipow_cases():
        a = cases.ipow_input(c)
        out = ref.array_scalar(orc.POW, a, c["exp"])
        _put(st, c["id"], out, (a,))
    np.savez_compressed(os.path.join(HERE, "ipow.npz"), **st)
    print("ipow:", len(st) // 4, "cases")

    st = {}
    for c in cases.dot_cases():
        a, b = cases.dot_inputs(c)
        out = np.array([ref.dot(a, b)], dtype=a.dtype)
        _put(st, c["id"], out, (a, b))
    np.savez_compressed(os.path.join(HERE, "dot.npz"), **st)
    print("dot:", len(st) // 4, "cases")

    make_dot_extra(ref)
    make_bench_sizes(ref)

    st = {}
    for c in cases.powf_cases():
        a = cases.powf_input(c)
        libm = ref.pow_apply(a, np.float32(c["exp"]))
        with np.errstate(all="ignore"):
            exact = np.power(a.astype(np.float64), np.float64(np.float32(c["exp"]))).astype(np.float32)
        # a signalling NaN operand yields NaN even for x**0 and 1**y (IEEE 754-2008 9.2.1, as glibc
        # does); the float32 -> float64 conversion above quiets it and hides that
        snan = np.isnan(a) & ((a.view(np.uint32) & np.uint32(0x00400000)) == 0)
        exact[snan] = np.nan
        _put(st, c["id"], libm, (a,))
        st[f"{c['id']}/exact"] = exact
    np.savez_compressed(os.path.join(HERE, "powf.npz"), **st)
    print("powf:", len(st) // 5, "cases")

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
