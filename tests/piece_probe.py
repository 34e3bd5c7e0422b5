"""Helper of test_gpu_parity.py::test_results_do_not_depend_on_the_piece_size: runs a fixed set of flat operations through
libsmhip and stores every result in an .npz.  The test runs it twice -- with the library's default launch rule (nothing at
these sizes is split) and with SMHIP_PIECE_LOG2VEC=14 (operands above 2^14 vectors go out in pieces of 2^14) -- and compares
the files bit for bit.       python tests/piece_probe.py out.npz"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import simplemath_amd as sma

lib = sma.load()
lib.set_device(0)
res = {}
user = lib.register_op("(a + b) * 2")
for n in ((1 << 14) * 4 * 3 + 5, (1 << 14) * 4 * 2, (1 << 14) * 4 + 1, (1 << 14) * 2 * 5 + 1):
    a = lib.uniform_f32(n, 41, 0.01, 100.0)
    b = lib.uniform_f32(n, 42, 0.5, 3.0)
    tag = f"n{n}"
    res[tag + "/add"] = lib.contiguous(sma.OP_ADD, a, b).numpy()
    res[tag + "/div"] = lib.contiguous(sma.OP_DIV, a, b).numpy()
    res[tag + "/mul_s"] = lib.array_scalar(sma.OP_MUL, a, np.float32(1.7)).numpy()
    res[tag + "/pow_s"] = lib.array_scalar(sma.OP_POW, a, np.float32(2.5)).numpy()
    res[tag + "/pow_a"] = lib.contiguous(sma.OP_POW, a, b).numpy()
    res[tag + "/user"] = lib.contiguous(user, a, b).numpy()
    res[tag + "/user_s"] = lib.array_scalar(user, a, np.float32(3.0)).numpy()
    ha, hb = a.numpy(), b.numpy()
    d64a, d64b = lib.to_device(ha.astype(np.float64)), lib.to_device(hb.astype(np.float64))
    res[tag + "/pow64_s"] = lib.array_scalar(sma.OP_POW, d64a, np.float64(2.5)).numpy()    # the double-double product chain
    res[tag + "/pow64_g"] = lib.array_scalar(sma.OP_POW, d64a, np.float64(2.7)).numpy()    # the general form
    res[tag + "/pow64_sq"] = lib.array_scalar(sma.OP_POW, d64a, np.float64(2.0)).numpy()   # a single IEEE operation
    res[tag + "/pow_rsqrt"] = lib.array_scalar(sma.OP_POW, d64a, np.float64(-0.5)).numpy()
    res[tag + "/add64"] = lib.contiguous(sma.OP_ADD, d64a, d64b).numpy()
    ia = lib.to_device((ha * 1000).astype(np.int32))
    ib = lib.to_device((hb * 1000).astype(np.int32))
    res[tag + "/imul"] = lib.contiguous(sma.OP_MUL, ia, ib).numpy()
    res[tag + "/sum"] = np.array([lib.sum(a)])
    res[tag + "/dot"] = np.array([lib.dot(a, b)])
    res[tag + "/idot"] = np.array([lib.dot(ia, ib)])
    out = lib.empty((n,), np.float32)
    sp = lib.alloc(8)
    lib.contiguous_sum_async(sma.OP_ADD, a, b, out, sp)
    res[tag + "/fused_sum"] = np.array([lib.read_f64(sp)])
    res[tag + "/fused_out"] = out.numpy()
    lib.free(sp)
    res[tag + "/expr_sum"] = np.array([lib.fused_expr_sum("(a0 - a1) * (a0 - a1)", a, b)])
np.savez(sys.argv[1], **res)
print("piece_probe ok", len(res))
