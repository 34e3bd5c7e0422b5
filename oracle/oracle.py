"""ctypes doorway to the CPU checker.  TEST INFRASTRUCTURE ONLY.

`Oracle()` loads oracle/libsmoracle.so (the C restatement, sm_oracle.c);
`Reference()` loads oracle/_ref/libsmref.so (the real reference headers behind
ref_shim.cpp) when it has been built.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path
(simplemath_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsmoracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsmref.so")

ADD, SUB, MUL, DIV, POW = range(5)
OPS = {"add": ADD, "sub": SUB, "mul": MUL, "div": DIV, "pow": POW}
F32, F64, I32, I64 = range(4)
DTYPES = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
          np.dtype(np.int32): I32, np.dtype(np.int64): I64}
# the generic dot_product<T>'s extra integer element types (smhip.h: SMHIP_I8 ... SMHIP_U64)
INT_KINDS = {np.dtype(np.int8): 4, np.dtype(np.uint8): 5, np.dtype(np.int16): 6, np.dtype(np.uint16): 7,
             np.dtype(np.uint32): 8, np.dtype(np.uint64): 9}
MAX_NDIM = 6

_szp = C.POINTER(C.c_size_t)


def build(ref: bool = True) -> None:
    """Compile the checker (and the reference shim where /root/reference exists)."""
    targets = ["all"] + (["ref"] if ref and os.path.isdir("/root/reference/include") else [])
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


def _sz(seq):
    arr = (C.c_size_t * max(len(seq), 1))(*[int(x) for x in seq])
    return arr


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def elem_strides(a: np.ndarray):
    """numpy byte strides -> the reference's element strides."""
    return [s // a.itemsize for s in a.strides]


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built (run `make -C oracle`)")
        self.lib = C.CDLL(path)


class Oracle(_Lib):
    """The C restatement (sm_oracle.c)."""

    def __init__(self):
        super().__init__(ORACLE_SO)
        L = self.lib
        L.smo_broadcast.restype = C.c_int
        L.smo_is_contiguous.restype = C.c_int
        L.smo_elementwise.restype = C.c_int
        L.smo_contiguous.restype = C.c_int
        L.smo_contiguous_mt.restype = C.c_int
        L.smo_array_scalar_mt.restype = C.c_int
        L.smo_contiguous_sum_mt.restype = C.c_double
        L.smo_set_threads.restype = None
        L.smo_array_scalar.restype = C.c_int
        L.smo_powi32.restype = C.c_int32
        L.smo_powi32.argtypes = [C.c_int32, C.c_int32]
        L.smo_dot.restype = C.c_int
        L.smo_sum_f64acc.restype = C.c_double
        L.smo_contiguous_sum.restype = C.c_double
        L.smo_uniform_f32.restype = C.c_float
        L.smo_uniform_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_float, C.c_float]
        L.smo_fill_uniform_f32.restype = None
        L.smo_num_threads.restype = C.c_int

    # -- shapes ---------------------------------------------------------
    def broadcast(self, shape1, strides1, shape2, strides2):
        """-> (result_shape, new_strides1, new_strides2, total) or None if incompatible."""
        nd = max(len(shape1), len(shape2))
        rs, s1, s2 = _sz([0] * nd), _sz([0] * nd), _sz([0] * nd)
        tot = C.c_size_t(0)
        rc = self.lib.smo_broadcast(C.c_int(len(shape1)), _sz(shape1), _sz(strides1),
                                    C.c_int(len(shape2)), _sz(shape2), _sz(strides2),
                                    rs, s1, s2, C.byref(tot))
        if rc < 0:
            return None
        return list(rs[:nd]), list(s1[:nd]), list(s2[:nd]), tot.value

    def is_contiguous(self, shape, stride):
        return bool(self.lib.smo_is_contiguous(C.c_int(len(shape)), _sz(shape), _sz(stride)))

    # -- loops ----------------------------------------------------------
    def elementwise(self, op, a, sa, b, sb, shape, quirk_1d=False):
        """a, b: flat base buffers (np arrays); sa/sb/shape in elements."""
        dt = DTYPES[a.dtype]
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        out = np.empty(n, dtype=a.dtype)
        rc = self.lib.smo_elementwise(C.c_int(op), C.c_int(dt), _ptr(a), _sz(sa), _ptr(b), _sz(sb),
                                      C.c_size_t(n), _ptr(out), _sz(shape), C.c_int(len(shape)),
                                      C.c_int(int(quirk_1d)))
        if rc:
            raise ValueError(f"smo_elementwise rc={rc}")
        return out

    def binary(self, op, a: np.ndarray, b: np.ndarray):
        """Broadcasted a op b for arbitrary (possibly strided) numpy views.

        Drives smo_broadcast + smo_elementwise exactly as SMArray::operator+
        does (SMArray.h:217-225).  Views must have non-negative strides.
        """
        assert a.dtype == b.dtype
        res = self.broadcast(a.shape, elem_strides(a), b.shape, elem_strides(b))
        if res is None:
            raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
        shape, sa, sb, _ = res
        abase, aoff = _base_and_offset(a)
        bbase, boff = _base_and_offset(b)
        out = self.elementwise(op, abase[aoff:], sa, bbase[boff:], sb, shape)
        return out.reshape(shape)

    def contiguous(self, op, a, b, out=None, mt=False):
        dt = DTYPES[a.dtype]
        if out is None:
            out = np.empty_like(a)
        fn = self.lib.smo_contiguous_mt if mt else self.lib.smo_contiguous
        rc = fn(C.c_int(op), C.c_int(dt), _ptr(a), _ptr(b), _ptr(out), C.c_size_t(a.size))
        if rc:
            raise ValueError(f"smo_contiguous rc={rc}")
        return out

    def array_scalar(self, op, a, value, int_pow_tail_libm=False, out=None):
        dt = DTYPES[a.dtype]
        v = np.array([value], dtype=a.dtype)
        if out is None:
            out = np.empty(a.size, dtype=a.dtype)
        rc = self.lib.smo_array_scalar(C.c_int(op), C.c_int(dt), _ptr(a), _ptr(v), C.c_size_t(a.size),
                                       _ptr(out), C.c_int(int(int_pow_tail_libm)))
        if rc:
            raise ValueError(f"smo_array_scalar rc={rc}")
        return out

    def powi32(self, base, exponent):
        return int(self.lib.smo_powi32(int(base), int(exponent)))

    def dot_c64(self, a, b, avx_body=False):
        """dot_product<std::complex<double>> restated: the definition (every element through the scalar statement), or as
        shipped (avx_body: the AVX body's doubled sums first).  a, b: complex128 arrays."""
        a, b = np.ascontiguousarray(a, dtype=np.complex128), np.ascontiguousarray(b, dtype=np.complex128)
        out = np.zeros(2, dtype=np.float64)
        self.lib.smo_dot_c64.restype = C.c_int
        rc = self.lib.smo_dot_c64(_ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out), C.c_int(1 if avx_body else 0))
        if rc:
            raise ValueError(f"smo_dot_c64 rc={rc}")
        return complex(out[0], out[1])

    def dot_c32(self, a, b, as_shipped=False):
        """The generic dot_product<std::complex<float>>: the reference's bits (as_shipped) or exact products with fp64 sums."""
        a, b = np.ascontiguousarray(a, dtype=np.complex64), np.ascontiguousarray(b, dtype=np.complex64)
        out = np.zeros(2, dtype=np.float32)
        self.lib.smo_dot_c32.restype = C.c_int
        rc = self.lib.smo_dot_c32(_ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out), C.c_int(1 if as_shipped else 0))
        if rc:
            raise ValueError(f"smo_dot_c32 rc={rc}")
        return np.complex64(complex(out[0], out[1]))

    def dot_int(self, a, b):
        """The generic dot_product<T> for int8/uint8/int16/uint16/uint32/uint64 arrays."""
        kind = INT_KINDS[a.dtype]
        out = np.zeros(1, dtype=a.dtype)
        self.lib.smo_dot_int.restype = C.c_int
        rc = self.lib.smo_dot_int(C.c_int(kind), _ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"smo_dot_int rc={rc}")
        return out[0]

    def dot(self, a, b, lane_order=False):
        dt = DTYPES[a.dtype]
        out = np.zeros(1, dtype=a.dtype)
        rc = self.lib.smo_dot(C.c_int(dt), _ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out),
                              C.c_int(int(lane_order)))
        if rc:
            raise ValueError(f"smo_dot rc={rc}")
        return out[0]

    def sum(self, a):
        return float(self.lib.smo_sum_f64acc(C.c_int(DTYPES[a.dtype]), _ptr(a), C.c_size_t(a.size)))

    def contiguous_sum(self, op, a, b):
        out = np.empty_like(a)
        s = self.lib.smo_contiguous_sum(C.c_int(op), C.c_int(DTYPES[a.dtype]), _ptr(a), _ptr(b),
                                        _ptr(out), C.c_size_t(a.size))
        return out, float(s)

    # -- synthetic inputs -------------------------------------------------
    def uniform_f32(self, n, seed, lo, hi, first=0):
        out = np.empty(n, dtype=np.float32)
        self.lib.smo_fill_uniform_f32(_ptr(out), C.c_size_t(n), C.c_uint64(seed), C.c_uint64(first),
                                      C.c_float(lo), C.c_float(hi))
        return out

    def num_threads(self):
        return int(self.lib.smo_num_threads())

    # -- all-core forms (bench.py's CPU baseline only) ----------------------
    def set_threads(self, n):
        self.lib.smo_set_threads(C.c_int(int(n)))

    def array_scalar_mt(self, op, a, value, out=None):
        v = np.array([value], dtype=a.dtype)
        if out is None:
            out = np.empty(a.size, dtype=a.dtype)
        rc = self.lib.smo_array_scalar_mt(C.c_int(op), C.c_int(DTYPES[a.dtype]), _ptr(a), _ptr(v), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"smo_array_scalar_mt rc={rc}")
        return out

    def contiguous_sum_mt(self, op, a, b, out=None):
        if out is None:
            out = np.empty_like(a)
        s = self.lib.smo_contiguous_sum_mt(C.c_int(op), C.c_int(DTYPES[a.dtype]), _ptr(a), _ptr(b), _ptr(out), C.c_size_t(a.size))
        return out, float(s)


class Reference(_Lib):
    """The real reference templates behind ref_shim.cpp (oracle/_ref/libsmref.so)."""

    def __init__(self):
        super().__init__(REF_SO)
        L = self.lib
        for name in ("ref_broadcast", "ref_is_contiguous", "ref_elementwise", "ref_array_scalar",
                     "ref_pow_apply", "ref_dot", "ref_smarray_binary", "ref_view_broadcast4"):
            getattr(L, name).restype = C.c_int
        L.ref_bench_add_f32.restype = C.c_double

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def broadcast(self, shape1, strides1, shape2, strides2):
        nd = max(len(shape1), len(shape2))
        rs, s1, s2 = _sz([0] * nd), _sz([0] * nd), _sz([0] * nd)
        tot = C.c_size_t(0)
        rc = self.lib.ref_broadcast(C.c_int(len(shape1)), _sz(shape1), _sz(strides1),
                                    C.c_int(len(shape2)), _sz(shape2), _sz(strides2),
                                    rs, s1, s2, C.byref(tot))
        if rc < 0:
            return None
        return list(rs[:nd]), list(s1[:nd]), list(s2[:nd]), tot.value

    def is_contiguous(self, shape, stride):
        return bool(self.lib.ref_is_contiguous(C.c_int(len(shape)), _sz(shape), _sz(stride)))

    def elementwise(self, op, a, sa, b, sb, shape):
        dt = DTYPES[a.dtype]
        n = int(np.prod(shape, dtype=np.int64))
        out = np.empty(n, dtype=a.dtype)
        rc = self.lib.ref_elementwise(C.c_int(op), C.c_int(dt), _ptr(a), _sz(sa), _ptr(b), _sz(sb),
                                      C.c_size_t(n), _ptr(out), _sz(shape), C.c_int(len(shape)))
        if rc:
            raise ValueError(f"ref_elementwise rc={rc}")
        return out

    def array_scalar(self, op, a, value):
        dt = DTYPES[a.dtype]
        v = np.array([value], dtype=a.dtype)
        out = np.empty(a.size, dtype=a.dtype)
        rc = self.lib.ref_array_scalar(C.c_int(op), C.c_int(dt), _ptr(a), _ptr(v), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"ref_array_scalar rc={rc}")
        return out

    def pow_apply(self, a, value):
        v = np.array([value], dtype=a.dtype)
        out = np.empty(a.size, dtype=a.dtype)
        rc = self.lib.ref_pow_apply(C.c_int(DTYPES[a.dtype]), _ptr(a), _ptr(v), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"ref_pow_apply rc={rc}")
        return out

    def dot_c64(self, a, b):
        a, b = np.ascontiguousarray(a, dtype=np.complex128), np.ascontiguousarray(b, dtype=np.complex128)
        out = np.zeros(2, dtype=np.float64)
        self.lib.ref_dot_c64.restype = C.c_int
        rc = self.lib.ref_dot_c64(_ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"ref_dot_c64 rc={rc}")
        return complex(out[0], out[1])

    def dot_c32(self, a, b):
        a, b = np.ascontiguousarray(a, dtype=np.complex64), np.ascontiguousarray(b, dtype=np.complex64)
        out = np.zeros(2, dtype=np.float32)
        self.lib.ref_dot_c32.restype = C.c_int
        rc = self.lib.ref_dot_c32(_ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"ref_dot_c32 rc={rc}")
        return np.complex64(complex(out[0], out[1]))

    def dot_int(self, a, b):
        out = np.zeros(1, dtype=a.dtype)
        self.lib.ref_dot_int.restype = C.c_int
        rc = self.lib.ref_dot_int(C.c_int(INT_KINDS[a.dtype]), _ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"ref_dot_int rc={rc}")
        return out[0]

    def dot(self, a, b):
        out = np.zeros(1, dtype=a.dtype)
        rc = self.lib.ref_dot(C.c_int(DTYPES[a.dtype]), _ptr(a), _ptr(b), C.c_size_t(a.size), _ptr(out))
        if rc:
            raise ValueError(f"ref_dot rc={rc}")
        return out[0]

    def smarray_binary(self, op, a, b, at=False, bt=False):
        """Dense arrays a, b through SMArray operators; at/bt view them via transpose()."""
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        ashape = a.shape[::-1] if at else a.shape
        bshape = b.shape[::-1] if bt else b.shape
        oshape_np = np.broadcast_shapes(ashape, bshape) if _compatible(ashape, bshape) else None
        n = int(np.prod(oshape_np)) if oshape_np is not None else 1
        out = np.empty(n, dtype=a.dtype)
        oshape = _sz([0] * MAX_NDIM)
        ond = C.c_int(0)
        rc = self.lib.ref_smarray_binary(C.c_int(op), C.c_int(DTYPES[a.dtype]), _ptr(a), _sz(a.shape),
                                         C.c_int(a.ndim), C.c_int(int(at)), _ptr(b), _sz(b.shape),
                                         C.c_int(b.ndim), C.c_int(int(bt)), _ptr(out), oshape, C.byref(ond))
        if rc == -1:
            raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
        if rc:
            raise ValueError(f"ref_smarray_binary rc={rc}")
        return out.reshape(list(oshape[:ond.value]))

    def view_broadcast4(self, op, big, small):
        big = np.ascontiguousarray(big, dtype=np.float32)
        small = np.ascontiguousarray(small, dtype=np.float32)
        d0, d1, d2, d3 = big.shape
        out = np.empty(d1 * d2 * d3, dtype=np.float32)
        oshape = _sz([0] * MAX_NDIM)
        nd = self.lib.ref_view_broadcast4(C.c_int(op), _ptr(big), _sz(big.shape), _ptr(small), _ptr(out), oshape)
        if nd < 0:
            raise ValueError("ref_view_broadcast4 failed")
        return out.reshape(list(oshape[:nd]))

    def bench_tiny(self, which, iters=200000):
        """ns per iteration of the reference's simple_check (0) / BM_SMArrayPow_1D (1) / BM_SMArrayPow_2D (2) body on this host."""
        self.lib.ref_bench_tiny.restype = C.c_double
        return float(self.lib.ref_bench_tiny(C.c_int(which), C.c_long(iters)))

    def bench_add_f32(self, a, b):
        return float(self.lib.ref_bench_add_f32(_ptr(a), _ptr(b), C.c_size_t(a.size)))


def _compatible(s1, s2):
    for d1, d2 in zip(s1[::-1], s2[::-1]):
        if d1 != d2 and d1 != 1 and d2 != 1:
            return False
    return True


def _base_and_offset(a: np.ndarray):
    """Flat owning buffer of a view + the view's element offset into it."""
    base = a
    while base.base is not None and isinstance(base.base, np.ndarray):
        base = base.base
    flat = base.reshape(-1) if base.flags.c_contiguous else np.ascontiguousarray(base).reshape(-1)
    if not base.flags.c_contiguous:
        # cannot alias: materialise the view densely instead
        dense = np.ascontiguousarray(a)
        return dense.reshape(-1), 0
    off = (a.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // a.itemsize
    return flat, int(off)


def ulp_diff_f32(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Distance in units-in-the-last-place between float32 arrays (NaN==NaN -> 0)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    xi = x.view(np.int32).astype(np.int64)
    yi = y.view(np.int32).astype(np.int64)
    xi = np.where(xi < 0, -(xi & 0x7FFFFFFF), xi)
    yi = np.where(yi < 0, -(yi & 0x7FFFFFFF), yi)
    d = np.abs(xi - yi)
    both_nan = np.isnan(x) & np.isnan(y)
    one_nan = np.isnan(x) ^ np.isnan(y)
    d = np.where(both_nan, 0, d)
    d = np.where(one_nan, np.iinfo(np.int64).max, d)
    return d


def ulp_diff_f64(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    xi = x.view(np.int64)
    yi = y.view(np.int64)
    sign = np.int64(-0x8000000000000000)
    xo = np.where(xi < 0, sign - xi, xi).astype(np.float64)  # monotone map; float64 is enough for small distances
    yo = np.where(yi < 0, sign - yi, yi).astype(np.float64)
    # exact for nearby values: subtract in integer space where it cannot overflow
    xm = np.where(xi < 0, -(xi & np.int64(0x7FFFFFFFFFFFFFFF)), xi)
    ym = np.where(yi < 0, -(yi & np.int64(0x7FFFFFFFFFFFFFFF)), yi)
    same_side = (np.abs(xo - yo) < 2.0 ** 62)
    d = np.where(same_side, np.abs(xm - ym), np.iinfo(np.int64).max)
    both_nan = np.isnan(x) & np.isnan(y)
    one_nan = np.isnan(x) ^ np.isnan(y)
    d = np.where(both_nan, 0, d)
    d = np.where(one_nan, np.iinfo(np.int64).max, d)
    return d
