"""Comparison helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_kat():
    with open(os.path.join(GOLDEN, "reference_kat.json")) as f:
        return json.load(f)["cases"]


def same_bits(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Elementwise: identical bit pattern, or both NaN (payload/sign of a NaN
    result is not part of the parity bar: x86 propagates an operand's payload,
    the GPU returns the canonical quiet NaN)."""
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y)
    assert x.dtype == y.dtype and x.shape == y.shape, (x.dtype, y.dtype, x.shape, y.shape)
    if x.dtype.kind == "f":
        u = {4: np.uint32, 8: np.uint64}[x.dtype.itemsize]
        return (x.view(u) == y.view(u)) | (np.isnan(x) & np.isnan(y))
    return x == y


def assert_same_bits(x, y, what=""):
    ok = same_bits(x, y)
    if not ok.all():
        bad = np.flatnonzero(~ok.reshape(-1))
        i = bad[0]
        raise AssertionError(f"{what}: {bad.size} of {ok.size} elements differ; first at {i}: "
                             f"{x.reshape(-1)[i]!r} vs {y.reshape(-1)[i]!r}")


def kat_array(spec, dtype):
    """Materialise a reference_kat.json array spec -> (dense base, view)."""
    if isinstance(spec, list):
        a = np.array(spec, dtype=dtype)
        return a, a
    shape = spec["shape"]
    if "fill" in spec:
        a = np.full(shape, spec["fill"], dtype=dtype)
        if "times_scalar" in spec:
            a = (a * dtype(spec["times_scalar"])).astype(dtype)
    else:
        n = int(np.prod(shape))
        even, odd = spec["alternate"]
        a = np.where(np.arange(n) % 2 == 0, even, odd).astype(dtype).reshape(shape)
    v = a
    for op in spec.get("view", []):
        assert op[0] == "index"
        v = v[op[1]]
    return a, v
