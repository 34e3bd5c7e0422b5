"""simplemath_amd -- MI355X (gfx950) implementation of simpleMath's element_wise_op hot path.

The product is `libsmhip.so` (HIP kernels behind the C ABI of include/smhip.h) plus the
header-only C++20 host side in include/ (sm.h, SMArray.h, ...).  This Python package is
only the thin ctypes binding the tests and bench.py drive the C ABI through -- the same
entry points the C++ headers call.  There is no CPU fallback anywhere in it: if the
library or a GPU is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import sys

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_PATH = os.path.join(PKG, "lib", "libsmhip.so")
HEADER = os.path.join(ROOT, "include", "smhip.h")

OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_POW, OP_LEFT = range(6)
OPS = {"add": OP_ADD, "sub": OP_SUB, "mul": OP_MUL, "div": OP_DIV, "pow": OP_POW, "left": OP_LEFT}
F32, F64, I32, I64 = range(4)
DTYPES = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.int32): I32, np.dtype(np.int64): I64}
# smhip_dot also takes the generic dot_product<T>'s other integer element types (smhip.h: SMHIP_I8 ... SMHIP_U64)
I8, U8, I16, U16, U32, U64 = range(4, 10)
DOT_DTYPES = {**DTYPES, np.dtype(np.int8): I8, np.dtype(np.uint8): U8, np.dtype(np.int16): I16, np.dtype(np.uint16): U16,
              np.dtype(np.uint32): U32, np.dtype(np.uint64): U64}
MAX_NDIM = 6

ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_BROADCAST = -1, -2, -3, -4, -5


class SmhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"smhip error {code}: {msg}")
        self.code = code


def declared_symbols():
    """Every function name include/smhip.h declares."""
    with open(HEADER) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(smhip_[a-z0-9_]+)\s*\(", text)))


def _i64(seq):
    return (C.c_int64 * max(len(seq), 1))(*[int(x) for x in seq])


class DeviceArray:
    """A (possibly strided) view on device memory: what sm::SMArray<T> holds host-side."""

    def __init__(self, lib, base_ptr, dtype, shape, strides, offset=0, owner=None):
        self.lib, self.base_ptr, self.dtype = lib, base_ptr, np.dtype(dtype)
        self.shape, self.strides, self.offset = tuple(int(s) for s in shape), tuple(int(s) for s in strides), int(offset)
        self._owner = owner if owner is not None else _Owner(lib, base_ptr)

    @property
    def ptr(self):
        # `lib.to_device(x).ptr` hands out the address of a block that returns to the pool the moment the expression ends
        # (two of round 3's red runs were that).  A temporary has no reference beyond the ones this call itself holds:
        # the evaluation stack, `self`, and getrefcount's argument.
        if sys.getrefcount(self) <= 3 and sys.getrefcount(self._owner) <= 2:  # (a temporary VIEW of a live array is fine: the owner lives on)
            raise RuntimeError("DeviceArray.ptr on a temporary: its memory goes back to the pool when this expression ends -- "
                               "keep the array in a variable for as long as the pointer is in use")
        return self.base_ptr + self.offset * self.dtype.itemsize

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    @property
    def ndim(self):
        return len(self.shape)

    def is_dense(self):
        exp = 1
        for d, s in zip(self.shape[::-1], self.strides[::-1]):
            if s != exp:
                return False
            exp *= d
        return True

    def view_like(self, np_view, np_base):
        """The same view numpy made on the host base, on the device base."""
        if np_view.size and not np.shares_memory(np_view, np_base):  # a view of ANOTHER array would turn into a wild device pointer
            raise ValueError("view_like: np_view is not a view of np_base")
        off = (np_view.__array_interface__["data"][0] - np_base.__array_interface__["data"][0]) // self.dtype.itemsize
        return DeviceArray(self.lib, self.base_ptr, self.dtype, np_view.shape,
                           [s // self.dtype.itemsize for s in np_view.strides], off, self._owner)

    def numpy(self):
        assert self.is_dense(), "download a dense array (views alias their parent)"
        out = np.empty(self.shape, dtype=self.dtype)
        self.lib.download(out, self.ptr)
        return out


class _Owner:
    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.free(self.ptr)
        except Exception:
            pass


class Smhip:
    """ctypes face of include/smhip.h."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: build it with `python -m simplemath_amd.build` "
                                    "(there is no CPU fallback)")
        self.path = path
        self.c = C.CDLL(path)
        c = self.c
        c.smhip_version.restype = C.c_char_p
        c.smhip_last_error.restype = C.c_char_p
        for name in declared_symbols():
            fn = getattr(c, name)  # AttributeError here = header/library mismatch
            if name not in ("smhip_version", "smhip_last_error"):
                fn.restype = C.c_int
        c.smhip_fill_uniform_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_float, C.c_float]

    # -- plumbing ---------------------------------------------------------
    def _ck(self, rc):
        if rc < 0:
            raise SmhipError(rc, self.c.smhip_last_error().decode())
        return rc

    def version(self):
        return self.c.smhip_version().decode()

    def device_count(self):
        n = C.c_int(0)
        self._ck(self.c.smhip_device_count(C.byref(n)))
        return n.value

    def set_device(self, d):
        self._ck(self.c.smhip_set_device(C.c_int(d)))

    def set_stream(self, stream_ptr):
        self._ck(self.c.smhip_set_stream(C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._ck(self.c.smhip_synchronize())

    def alloc(self, nbytes):
        p = C.c_void_p(0)
        self._ck(self.c.smhip_alloc(C.byref(p), C.c_size_t(nbytes)))
        return p.value

    def free(self, ptr):
        self._ck(self.c.smhip_free(C.c_void_p(ptr)))

    def pool_trim(self):
        self._ck(self.c.smhip_pool_trim())

    def pool_stats(self):
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.c.smhip_pool_stats(C.byref(a), C.byref(b)))
        return a.value, b.value

    def upload(self, ptr, host: np.ndarray):
        host = np.ascontiguousarray(host)
        self._ck(self.c.smhip_upload(C.c_void_p(ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes)))

    def copy(self, dst_ptr, src_ptr, nbytes):
        """Device-to-device copy on the library's stream."""
        self._ck(self.c.smhip_copy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes)))

    def download(self, host: np.ndarray, ptr):
        assert host.flags.c_contiguous
        self._ck(self.c.smhip_download(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(host.nbytes)))

    # -- arrays -----------------------------------------------------------
    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in np.atleast_1d(shape))
        n = int(np.prod(shape, dtype=np.int64))
        ptr = self.alloc(max(n, 1) * dtype.itemsize)
        strides, acc = [], 1
        for d in shape[::-1]:
            strides.append(acc)
            acc *= d
        return DeviceArray(self, ptr, dtype, shape, strides[::-1])

    def to_device(self, host: np.ndarray):
        host = np.ascontiguousarray(host)
        d = self.empty(host.shape if host.ndim else (1,), host.dtype)
        self.upload(d.ptr, host)
        return d

    def full(self, shape, value, dtype):
        d = self.empty(shape, dtype)
        v = np.array([value], dtype=dtype)
        self._ck(self.c.smhip_fill(C.c_int(DTYPES[d.dtype]), C.c_void_p(d.ptr), v.ctypes.data_as(C.c_void_p), C.c_size_t(d.size)))
        return d

    def uniform_f32(self, n, seed, lo, hi, first=0):
        d = self.empty((n,), np.float32)
        self._ck(self.c.smhip_fill_uniform_f32(C.c_void_p(d.ptr), n, seed, first, lo, hi))
        return d

    # -- shape layer --------------------------------------------------------
    def broadcast(self, shape1, strides1, shape2, strides2):
        nd = max(len(shape1), len(shape2))
        rs, s1, s2 = _i64([0] * nd), _i64([0] * nd), _i64([0] * nd)
        tot = C.c_int64(0)
        rc = self.c.smhip_broadcast(C.c_int(len(shape1)), _i64(shape1), _i64(strides1), C.c_int(len(shape2)),
                                    _i64(shape2), _i64(strides2), rs, s1, s2, C.byref(tot))
        if rc == ERR_BROADCAST:
            return None
        self._ck(rc)
        return list(rs[:nd]), list(s1[:nd]), list(s2[:nd]), tot.value

    def is_contiguous(self, shape, strides):
        return bool(self.c.smhip_is_contiguous(C.c_int(len(shape)), _i64(shape), _i64(strides)))

    def register_op(self, hip_expression: str) -> int:
        """User-defined Op: a HIP expression in `a` and `b`; returns the op id to pass as `op`."""
        oid = C.c_int(0)
        self._ck(self.c.smhip_register_op(hip_expression.encode(), C.byref(oid)))
        return oid.value

    # -- hot path -------------------------------------------------------------
    def elementwise_raw(self, op, dtype, a_ptr, sa, b_ptr, sb, shape, out_ptr):
        self._ck(self.c.smhip_elementwise(C.c_int(op), C.c_int(DTYPES[np.dtype(dtype)]), C.c_void_p(a_ptr), _i64(sa),
                                          C.c_void_p(b_ptr), _i64(sb), _i64(shape), C.c_int(len(shape)), C.c_void_p(out_ptr)))

    def binary(self, op, a: DeviceArray, b: DeviceArray, out: DeviceArray | None = None):
        """a op b with broadcasting: what SMArray::operator+ does (SMArray.h:217-225)."""
        assert a.dtype == b.dtype
        res = self.broadcast(a.shape, a.strides, b.shape, b.strides)
        if res is None:
            raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
        shape, sa, sb, _ = res
        if out is None:
            out = self.empty(shape, a.dtype)
        self.elementwise_raw(op, a.dtype, a.ptr, sa, b.ptr, sb, shape, out.ptr)
        return out

    def binary_inline(self, op, a, b):
        """a op b where a host numpy array (<= 1 KiB) rides in the kernel's argument block instead of being uploaded
        (smhip_elementwise_inline); a, b: numpy arrays (inline) or DeviceArrays; a numpy scalar is an inline operand of one element."""
        def side(x):
            if isinstance(x, DeviceArray):
                return x.shape, x.strides, C.c_void_p(x.ptr), 0, x.dtype, None
            h = np.ascontiguousarray(x)
            if h.ndim == 0:
                h = h.reshape(1)
            return h.shape, [st // h.itemsize for st in h.strides], h.ctypes.data_as(C.c_void_p), h.nbytes, h.dtype, h
        sha, sta, pa, na, dta, keep_a = side(a)
        shb, stb, pb, nb, dtb, keep_b = side(b)
        assert dta == dtb
        res = self.broadcast(sha, sta, shb, stb)
        if res is None:
            raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
        shape, sa, sb, _ = res
        out = self.empty(shape, dta)
        self._ck(self.c.smhip_elementwise_inline(C.c_int(op), C.c_int(DTYPES[np.dtype(dta)]), pa, C.c_size_t(na), _i64(sa), pb, C.c_size_t(nb),
                                                 _i64(sb), _i64(shape), C.c_int(len(shape)), C.c_void_p(out.ptr)))
        return out

    def assign(self, dst: DeviceArray, src: DeviceArray):
        """dst[...] = src, element by element (src broadcast to dst's shape): SMArray::operator=(SMArray&&), SMArray.h:89-97."""
        assert dst.dtype == src.dtype
        res = self.broadcast(dst.shape, dst.strides, src.shape, src.strides)
        if res is None or tuple(res[0]) != tuple(dst.shape):
            raise RuntimeError("Shape mismatch in assignment")
        shape, sd, ss, _ = res
        self._ck(self.c.smhip_copy_strided(C.c_int(DTYPES[dst.dtype]), C.c_void_p(src.ptr), _i64(ss), C.c_void_p(dst.ptr), _i64(sd),
                                           _i64(shape), C.c_int(len(shape))))

    def contiguous(self, op, a: DeviceArray, b: DeviceArray, out: DeviceArray | None = None):
        assert a.dtype == b.dtype and a.size == b.size
        if out is None:
            out = self.empty(a.shape, a.dtype)
        self._ck(self.c.smhip_contiguous(C.c_int(op), C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr), C.c_void_p(b.ptr),
                                         C.c_void_p(out.ptr), C.c_size_t(a.size)))
        return out

    def array_scalar(self, op, a: DeviceArray, value, out: DeviceArray | None = None):
        if out is None:
            out = self.empty(a.shape, a.dtype)
        v = np.array([value], dtype=a.dtype)
        self._ck(self.c.smhip_array_scalar(C.c_int(op), C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr),
                                           v.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_void_p(out.ptr)))
        return out

    def fused(self, op1, op2, a: DeviceArray, b: DeviceArray, c, out: DeviceArray | None = None):
        """(a op1 b) op2 c in one pass; c a DeviceArray or a scalar."""
        if out is None:
            out = self.empty(a.shape, a.dtype)
        if isinstance(c, DeviceArray):
            cp, sp = C.c_void_p(c.ptr), C.c_void_p(0)
        else:
            v = np.array([c], dtype=a.dtype)
            cp, sp = C.c_void_p(0), v.ctypes.data_as(C.c_void_p)
        self._ck(self.c.smhip_fused_contiguous(C.c_int(op1), C.c_int(op2), C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr),
                                               C.c_void_p(b.ptr), cp, sp, C.c_void_p(out.ptr), C.c_size_t(a.size)))
        return out

    def chain_call(self, first: DeviceArray, *stages, out: DeviceArray | None = None):
        """(callable, out): the prepared smhip_chain call for `chain(first, *stages)` -- timing loops call it without
        paying the argument marshalling again."""
        dt = first.dtype
        shape = list(first.shape)
        for st in stages:
            x = st[1]
            if isinstance(x, DeviceArray):
                assert x.dtype == dt
                res = self.broadcast(shape, [0] * len(shape), x.shape, x.strides)
                if res is None:
                    raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
                shape = res[0]
        nd = len(shape)
        operands = [first] + [st[1] for st in stages]
        strides, ptrs, scal = [], [], np.zeros(len(operands), dtype=dt)
        for k, x in enumerate(operands):
            if isinstance(x, DeviceArray):
                res = self.broadcast(shape, [0] * nd, x.shape, x.strides)
                strides += list(res[2])
                ptrs.append(x.ptr)
            else:
                strides += [0] * nd
                ptrs.append(None)
                scal[k] = x
        if out is None:
            out = self.empty(shape, dt)
        ops = (C.c_int * len(stages))(*[int(st[0]) for st in stages])
        swp = (C.c_int * len(stages))(*[1 if len(st) > 2 and st[2] else 0 for st in stages])
        args = (C.c_int(DTYPES[dt]), C.c_int(len(operands)), (C.c_void_p * len(operands))(*ptrs), _i64(strides),
                scal.ctypes.data_as(C.c_void_p), ops, swp, _i64(shape), C.c_int(nd), C.c_void_p(out.ptr))
        keep = (operands, scal, out)  # the arrays stay alive as long as the callable does
        fn = self.c.smhip_chain

        def call(_keep=keep):
            rc = fn(*args)
            if rc < 0:
                self._ck(rc)
        return call, out

    def chain_sum_call(self, first: DeviceArray, *stages):
        """(callable -> float): the prepared smhip_chain_sum call for the sum of `chain(first, *stages)`'s value, which is not written."""
        dt = first.dtype
        shape = list(first.shape)
        for st in stages:
            x = st[1]
            if isinstance(x, DeviceArray):
                assert x.dtype == dt
                res = self.broadcast(shape, [0] * len(shape), x.shape, x.strides)
                if res is None:
                    raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
                shape = res[0]
        nd = len(shape)
        operands = [first] + [st[1] for st in stages]
        strides, ptrs, scal = [], [], np.zeros(len(operands), dtype=dt)
        for k, x in enumerate(operands):
            if isinstance(x, DeviceArray):
                res = self.broadcast(shape, [0] * nd, x.shape, x.strides)
                strides += list(res[2])
                ptrs.append(x.ptr)
            else:
                strides += [0] * nd
                ptrs.append(None)
                scal[k] = x
        ops = (C.c_int * len(stages))(*[int(st[0]) for st in stages])
        swp = (C.c_int * len(stages))(*[1 if len(st) > 2 and st[2] else 0 for st in stages])
        result = C.c_double(0.0)
        args = (C.c_int(DTYPES[dt]), C.c_int(len(operands)), (C.c_void_p * len(operands))(*ptrs), _i64(strides),
                scal.ctypes.data_as(C.c_void_p), ops, swp, _i64(shape), C.c_int(nd), C.byref(result))
        keep = (operands, scal)
        fn = self.c.smhip_chain_sum

        def call(_keep=keep):
            self._ck(fn(*args))
            return result.value
        return call

    def chain_sum(self, first: DeviceArray, *stages):
        """sum(chain(first, *stages)) without writing the chain's value (smhip_chain_sum): fp64 / wrapping 64-bit accumulation."""
        return self.chain_sum_call(first, *stages)()

    def chain(self, first: DeviceArray, *stages, out: DeviceArray | None = None):
        """An operator chain in as few passes as possible (smhip_chain): r = first; then for each stage (op, x) -- or
        (op, x, True) for the swapped form x op r -- r = r op x with NumPy broadcasting; x a DeviceArray or a scalar.
        What SMArray's operators queue when a temporary feeds the next operator of the same expression."""
        call, out = self.chain_call(first, *stages, out=out)
        call()
        return out

    def fused_expr(self, expression: str, *arrays: DeviceArray, scalars=(), out: DeviceArray | None = None):
        """out = EXPR(a0, a1, ..., s0, ...) in one pass over dense, equal-sized operands (smhip_fused_expr)."""
        a0 = arrays[0]
        assert all(a.dtype == a0.dtype and a.size == a0.size and a.is_dense() for a in arrays)
        if out is None:
            out = self.empty(a0.shape, a0.dtype)
        ptrs = (C.c_void_p * len(arrays))(*[a.ptr for a in arrays])
        sc = np.array(list(scalars), dtype=a0.dtype)
        self._ck(self.c.smhip_fused_expr(expression.encode(), C.c_int(DTYPES[a0.dtype]), ptrs, C.c_int(len(arrays)),
                                         sc.ctypes.data_as(C.c_void_p) if len(sc) else None, C.c_int(len(sc)), C.c_void_p(out.ptr),
                                         C.c_size_t(a0.size)))
        return out

    def fused_expr_bcast(self, expression: str, *arrays: DeviceArray, scalars=(), out: DeviceArray | None = None):
        """out = EXPR(a0, a1, ..., s0, ...) in one pass over operands that broadcast against each other (smhip_fused_expr_bcast)."""
        a0 = arrays[0]
        shape = list(a0.shape)
        for a in arrays[1:]:
            assert a.dtype == a0.dtype
            res = self.broadcast(shape, [0] * len(shape), a.shape, a.strides)
            if res is None:
                raise RuntimeError("Cannot broadcast shapes: incompatible dimensions")
            shape = res[0]
        nd = len(shape)
        strides = []
        for a in arrays:
            strides += list(self.broadcast(shape, [0] * nd, a.shape, a.strides)[2])
        if out is None:
            out = self.empty(shape, a0.dtype)
        ptrs = (C.c_void_p * len(arrays))(*[a.ptr for a in arrays])
        sc = np.array(list(scalars), dtype=a0.dtype)
        self._ck(self.c.smhip_fused_expr_bcast(expression.encode(), C.c_int(DTYPES[a0.dtype]), ptrs, _i64(strides), C.c_int(len(arrays)),
                                               sc.ctypes.data_as(C.c_void_p) if len(sc) else None, C.c_int(len(sc)), _i64(shape), C.c_int(nd),
                                               C.c_void_p(out.ptr)))
        return out

    def fused_expr_sum(self, expression: str, *arrays: DeviceArray, scalars=(), store=False):
        """sum_i EXPR(a0[i], ...) in one pass; store=True also returns the elementwise result.  Blocks for the value."""
        a0 = arrays[0]
        assert all(a.dtype == a0.dtype and a.size == a0.size and a.is_dense() for a in arrays)
        out = self.empty(a0.shape, a0.dtype) if store else None
        ptrs = (C.c_void_p * len(arrays))(*[a.ptr for a in arrays])
        sc = np.array(list(scalars), dtype=a0.dtype)
        sp = self.alloc(8)
        try:
            self._ck(self.c.smhip_fused_expr_sum_async(expression.encode(), C.c_int(DTYPES[a0.dtype]), ptrs, C.c_int(len(arrays)),
                                                       sc.ctypes.data_as(C.c_void_p) if len(sc) else None, C.c_int(len(sc)),
                                                       C.c_void_p(out.ptr) if store else None, C.c_size_t(a0.size), C.c_void_p(sp)))
            total = self.read_f64(sp)
        finally:
            self.free(sp)
        return (total, out) if store else total

    def dot(self, a: DeviceArray, b: DeviceArray):
        out = np.zeros(1, dtype=a.dtype)
        self._ck(self.c.smhip_dot(C.c_int(DOT_DTYPES[a.dtype]), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_size_t(a.size),
                                  out.ctypes.data_as(C.c_void_p)))
        return out[0]

    def dot_c64(self, a_ptr, b_ptr, n):
        out = np.zeros(2, dtype=np.float64)
        self._ck(self.c.smhip_dot_c64(C.c_void_p(a_ptr), C.c_void_p(b_ptr), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return complex(out[0], out[1])

    def dot_c32(self, a_ptr, b_ptr, n):
        out = np.zeros(2, dtype=np.float32)
        self._ck(self.c.smhip_dot_c32(C.c_void_p(a_ptr), C.c_void_p(b_ptr), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return np.complex64(complex(out[0], out[1]))

    def sum(self, a: DeviceArray):
        out = C.c_double(0)
        self._ck(self.c.smhip_sum(C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr), C.c_size_t(a.size), C.byref(out)))
        return out.value

    def sum_async(self, a: DeviceArray, out_ptr):
        self._ck(self.c.smhip_sum_async(C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr), C.c_size_t(a.size), C.c_void_p(out_ptr)))

    def dot_async(self, a: DeviceArray, b: DeviceArray, out_ptr):
        self._ck(self.c.smhip_dot_async(C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_size_t(a.size),
                                        C.c_void_p(out_ptr)))

    def dot_c64_async(self, a_ptr, b_ptr, n, out2_ptr):
        self._ck(self.c.smhip_dot_c64_async(C.c_void_p(a_ptr), C.c_void_p(b_ptr), C.c_size_t(n), C.c_void_p(out2_ptr)))

    def contiguous_sum_async(self, op, a: DeviceArray, b: DeviceArray, out: DeviceArray, sum_ptr):
        self._ck(self.c.smhip_contiguous_sum_async(C.c_int(op), C.c_int(DTYPES[a.dtype]), C.c_void_p(a.ptr), C.c_void_p(b.ptr),
                                                   C.c_void_p(out.ptr), C.c_size_t(a.size), C.c_void_p(sum_ptr)))

    def read_f64(self, ptr):
        out = np.zeros(1, dtype=np.float64)
        self.download(out, ptr)
        return float(out[0])

    def read_i64(self, ptr):
        out = np.zeros(1, dtype=np.int64)
        self.download(out, ptr)
        return int(out[0])

    # -- multi-GPU ----------------------------------------------------------------
    def split_range(self, n, world, rank):
        st, ct = C.c_int64(0), C.c_int64(0)
        self._ck(self.c.smhip_split_range(C.c_int64(n), C.c_int(world), C.c_int(rank), C.byref(st), C.byref(ct)))
        return st.value, ct.value

    def shard_outer(self, shape, strides_a, strides_b, world, rank):
        """The C planner behind sm::Sharded / smhip_sharded_elementwise; same answer as simplemath_amd.sharding.shard_outer."""
        nd = len(shape)
        local = _i64([0] * nd)
        oa, ob, oo = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        mask = C.c_int(0)
        self._ck(self.c.smhip_shard_outer(_i64(shape), _i64(strides_a), _i64(strides_b), C.c_int(nd), C.c_int(world), C.c_int(rank),
                                          local, C.byref(oa), C.byref(ob), C.byref(oo), C.byref(mask)))
        return tuple(local[:nd]), oa.value, ob.value, oo.value, bool(mask.value & 1), bool(mask.value & 2)

    def set_devices(self, n):
        self._ck(self.c.smhip_set_devices(C.c_int(n)))

    def get_devices(self):
        n = C.c_int(0)
        self._ck(self.c.smhip_get_devices(C.byref(n)))
        return n.value

    def group_info(self, index):
        """(nranks, rank, device) as RCCL reports them for the device group's communicator `index`."""
        n, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(self.c.smhip_group_info(C.c_int(index), C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value

    def rccl_version(self):
        v = C.c_int(0)
        self._ck(self.c.smhip_rccl_version(C.byref(v)))
        return v.value

    def copy_peer(self, dst_ptr, dst_device, src_ptr, src_device, nbytes):
        self._ck(self.c.smhip_copy_peer(C.c_void_p(dst_ptr), C.c_int(dst_device), C.c_void_p(src_ptr), C.c_int(src_device), C.c_size_t(nbytes)))

    def sharded_synchronize(self):
        self._ck(self.c.smhip_sharded_synchronize())

    @staticmethod
    def _ptr_table(ptrs):
        return (C.c_void_p * len(ptrs))(*ptrs)

    @staticmethod
    def _size_table(ns):
        return (C.c_size_t * len(ns))(*ns)

    def sharded_contiguous(self, op, dtype, a_ptrs, b_ptrs, out_ptrs, ns):
        self._ck(self.c.smhip_sharded_contiguous(C.c_int(op), C.c_int(DTYPES[np.dtype(dtype)]), self._ptr_table(a_ptrs),
                                                 self._ptr_table(b_ptrs), self._ptr_table(out_ptrs), self._size_table(ns)))

    def sharded_array_scalar(self, op, dtype, a_ptrs, value, ns, out_ptrs):
        v = np.array([value], dtype=dtype)
        self._ck(self.c.smhip_sharded_array_scalar(C.c_int(op), C.c_int(DTYPES[np.dtype(dtype)]), self._ptr_table(a_ptrs),
                                                   v.ctypes.data_as(C.c_void_p), self._size_table(ns), self._ptr_table(out_ptrs)))

    def sharded_elementwise(self, op, dtype, a_ptrs, sa, b_ptrs, sb, shape, out_ptrs):
        self._ck(self.c.smhip_sharded_elementwise(C.c_int(op), C.c_int(DTYPES[np.dtype(dtype)]), self._ptr_table(a_ptrs), _i64(sa),
                                                  self._ptr_table(b_ptrs), _i64(sb), _i64(shape), C.c_int(len(shape)),
                                                  self._ptr_table(out_ptrs)))

    def sharded_contiguous_sum(self, op, dtype, a_ptrs, b_ptrs, out_ptrs, ns):
        total = C.c_double(0)
        self._ck(self.c.smhip_sharded_contiguous_sum(C.c_int(op), C.c_int(DTYPES[np.dtype(dtype)]), self._ptr_table(a_ptrs),
                                                     self._ptr_table(b_ptrs), self._ptr_table(out_ptrs), self._size_table(ns),
                                                     C.byref(total)))
        return total.value

    def sharded_sum(self, dtype, a_ptrs, ns):
        total = C.c_double(0)
        self._ck(self.c.smhip_sharded_sum(C.c_int(DTYPES[np.dtype(dtype)]), self._ptr_table(a_ptrs), self._size_table(ns), C.byref(total)))
        return total.value

    def sharded_dot(self, dtype, a_ptrs, b_ptrs, ns):
        out = np.zeros(1, dtype=dtype)
        self._ck(self.c.smhip_sharded_dot(C.c_int(DTYPES[np.dtype(dtype)]), self._ptr_table(a_ptrs), self._ptr_table(b_ptrs),
                                          self._size_table(ns), out.ctypes.data_as(C.c_void_p)))
        return out[0]

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._ck(self.c.smhip_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, nranks, rank, unique_id: bytes):
        assert len(unique_id) == 128
        self._ck(self.c.smhip_comm_init_rank(C.c_int(nranks), C.c_int(rank), C.c_char_p(unique_id)))

    def comm_info(self):
        n, r = C.c_int(0), C.c_int(0)
        self._ck(self.c.smhip_comm_info(C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_destroy(self):
        self._ck(self.c.smhip_comm_destroy())

    def allreduce_sum_async(self, dtype, ptr, count=1):
        self._ck(self.c.smhip_allreduce_sum_async(C.c_int(DTYPES[np.dtype(dtype)]), C.c_void_p(ptr), C.c_size_t(count)))

    def policy_probe(self, a=0, a_bytes=0, b=0, b_bytes=0, out=0, out_bytes=0):
        """The stream-policy word for a launch with these operand spans (bit 0: nt reads, bit 1: keep-stores); records the touches."""
        pol = C.c_int(0)
        self._ck(self.c.smhip_policy_probe(C.c_void_p(a), C.c_size_t(a_bytes), C.c_void_p(b), C.c_size_t(b_bytes), C.c_void_p(out),
                                           C.c_size_t(out_bytes), C.byref(pol)))
        return pol.value

    def policy_peek(self, a=0, a_bytes=0, b=0, b_bytes=0, out=0, out_bytes=0):
        """policy_probe without recording the spans as touched."""
        pol = C.c_int(0)
        self._ck(self.c.smhip_policy_peek(C.c_void_p(a), C.c_size_t(a_bytes), C.c_void_p(b), C.c_size_t(b_bytes), C.c_void_p(out),
                                          C.c_size_t(out_bytes), C.byref(pol)))
        return pol.value

    def queue_stats(self):
        """(queues in use on this thread's device, operators that took the other queue, cross-queue event edges)"""
        q, alt, edges = C.c_int(0), C.c_ulonglong(0), C.c_ulonglong(0)
        self._ck(self.c.smhip_queue_stats(C.byref(q), C.byref(alt), C.byref(edges)))
        return q.value, alt.value, edges.value

    def tiny_stats(self):
        """(launches that carried recorded tiny operators, operators they carried) on this thread's device (csrc/tiny.hip)"""
        launches, ops = C.c_ulonglong(0), C.c_ulonglong(0)
        self._ck(self.c.smhip_tiny_stats(C.byref(launches), C.byref(ops)))
        return launches.value, ops.value

    def launch_pieces(self, bytes_per_operand, streams=3):
        """Kernel launches a dense streaming operator over operands of this size goes out as (streams: 3 a op b, 2 a op s / dot, 1 sum)."""
        k = C.c_int(0)
        self._ck(self.c.smhip_launch_pieces(C.c_size_t(bytes_per_operand), C.c_int(streams), C.byref(k)))
        return k.value

    # -- timing -----------------------------------------------------------------
    def event(self):
        e = C.c_void_p(0)
        self._ck(self.c.smhip_event_create(C.byref(e)))
        return e.value

    def record(self, ev):
        self._ck(self.c.smhip_event_record(C.c_void_p(ev)))

    def event_sync(self, ev):
        self._ck(self.c.smhip_event_synchronize(C.c_void_p(ev)))

    def elapsed_ms(self, e0, e1):
        ms = C.c_float(0)
        self._ck(self.c.smhip_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1), C.byref(ms)))
        return ms.value

    def event_destroy(self, ev):
        self._ck(self.c.smhip_event_destroy(C.c_void_p(ev)))


_lib = None


def load(path: str = LIB_PATH) -> Smhip:
    """The loaded library (cached).  Raises if libsmhip.so has not been built.  SMHIP_LIBRARY=<path> substitutes an
    experimental build (tools/build_variant.sh) for the default one."""
    global _lib
    if path == LIB_PATH and os.environ.get("SMHIP_LIBRARY"):
        path = os.environ["SMHIP_LIBRARY"]
    if _lib is None or _lib.path != path:
        _lib = Smhip(path)
    return _lib
