"""libsmhip's multi-GPU layer (include/smhip.h "multi-GPU", csrc/sharded.hip, include/Sharded.h).

CPU part: the C shard planner (smhip_split_range / smhip_shard_outer -- host-only code) gives the same blocks as
simplemath_amd/sharding.py, whose stitched results tests/test_sharding.py checks against the oracle over gloo with 2 and 3
ranks; the sharded entry points refuse loudly without a GPU.
GPU part (one-GPU box): the whole sharded path through the REAL RCCL calls with a one-device group / one-rank
communicator -- ncclCommInitAll, ncclGroupStart/End, ncclAllReduce, ncclCommInitRank -- checked against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import simplemath_amd as sma  # noqa: E402
from simplemath_amd import sharding  # noqa: E402


def test_c_planner_equals_python_planner(smhip):
    for n in (0, 1, 5, 8, 17, 4096, (1 << 31) + 3):
        for w in (1, 2, 3, 8):
            for r in range(w):
                assert smhip.split_range(n, w, r) == sharding.split_range(n, w, r)
    problems = [((4096, 4096), (4096, 1), (0, 1)),            # config 3: the row is replicated
                ((1 << 31,), (1,), (1,)),                      # config 5
                ((7, 3, 5), (15, 5, 1), (15, 0, 1)),
                ((1, 224, 224, 3), (0, 672, 3, 1), (0, 3, 0, 1)),  # the reference tests' 4-D pattern: dim 0 has one row
                ((2, 5), (5, 1), (5, 1)),                      # fewer rows than ranks
                ((5,), (1,), (0,))]
    for shape, sa, sb in problems:
        for w in (1, 2, 3, 4, 8):
            for r in range(w):
                s = sharding.shard_outer(shape, sa, sb, w, r)
                got = smhip.shard_outer(shape, sa, sb, w, r)
                assert got == (s.shape, s.offset_a, s.offset_b, s.offset_out, s.replicated_a, s.replicated_b), (shape, w, r)


def test_planner_rejects_bad_arguments(smhip):
    with pytest.raises(sma.SmhipError):
        smhip.split_range(10, 0, 0)
    with pytest.raises(sma.SmhipError):
        smhip.split_range(10, 2, 2)
    with pytest.raises(sma.SmhipError):
        smhip.shard_outer((1,) * 7, (1,) * 7, (1,) * 7, 2, 0)  # rank above MAX_NDIM


def test_sharded_entry_points_need_a_group(smhip):
    """Without smhip_set_devices every sharded call fails with a message that says what to do (and never computes)."""
    assert smhip.get_devices() == 0
    with pytest.raises(sma.SmhipError, match="smhip_set_devices"):
        smhip.sharded_contiguous(sma.OP_ADD, np.float32, [0], [0], [0], [0])
    with pytest.raises(sma.SmhipError, match="smhip_set_devices"):
        smhip.sharded_sum(np.float32, [0], [0])
    with pytest.raises(sma.SmhipError, match="smhip_comm_init_rank"):
        smhip.allreduce_sum_async(np.float64, 1, 1)  # no communicator (and, on CPU, no device either)


# --------------------------------------------------------------------------------------------- GPU

@pytest.fixture
def group1(smhip):
    smhip.set_device(0)
    smhip.set_devices(1)  # ncclCommInitAll with one device
    yield smhip
    smhip.set_devices(0)


@pytest.mark.gpu
def test_sharded_elementwise_and_sum_through_rccl(group1, oracle):
    from oracle import oracle as orc
    lib = group1
    assert lib.get_devices() == 1
    n = (1 << 22) + 3
    a, b = lib.uniform_f32(n, 6, 0.0, 1.0), lib.uniform_f32(n, 7, 0.0, 1.0)
    ha, hb = oracle.uniform_f32(n, 6, 0.0, 1.0), oracle.uniform_f32(n, 7, 0.0, 1.0)
    out = lib.empty((n,), np.float32)
    lib.sharded_contiguous(sma.OP_ADD, np.float32, [a.ptr], [b.ptr], [out.ptr], [n])
    lib.sharded_synchronize()
    want, want_sum = oracle.contiguous_sum(orc.ADD, ha, hb)
    assert np.array_equal(out.numpy(), want)
    # config 5: fused add + sum, then ncclGroupStart / ncclAllReduce(1 x fp64) / ncclGroupEnd
    out2 = lib.empty((n,), np.float32)
    total = lib.sharded_contiguous_sum(sma.OP_ADD, np.float32, [a.ptr], [b.ptr], [out2.ptr], [n])
    assert np.array_equal(out2.numpy(), want)
    assert abs(total - want_sum) <= 1e-12 * want_sum
    assert abs(lib.sharded_sum(np.float32, [out.ptr], [n]) - want_sum) <= 1e-12 * want_sum
    # dot: floats through fp64 partials, integers wrapping exactly
    assert abs(float(lib.sharded_dot(np.float32, [a.ptr], [b.ptr], [n])) - float(np.dot(ha.astype(np.float64), hb.astype(np.float64)))) <= 1e-6 * n
    rng = np.random.default_rng(5)
    ia = rng.integers(-2**31, 2**31 - 1, size=100003, dtype=np.int64).astype(np.int32)
    ib = rng.integers(-2**31, 2**31 - 1, size=100003, dtype=np.int64).astype(np.int32)
    da, db = lib.to_device(ia), lib.to_device(ib)
    assert int(lib.sharded_dot(np.int32, [da.ptr], [db.ptr], [ia.size])) == int(oracle.dot(ia, ib))
    # scalar operand
    o3 = lib.empty((n,), np.float32)
    lib.sharded_array_scalar(sma.OP_MUL, np.float32, [a.ptr], 3.0, [n], [o3.ptr])
    assert np.array_equal(o3.numpy(), ha * np.float32(3.0))


@pytest.mark.gpu
def test_sharded_broadcast_config3_shape(group1, oracle):
    from oracle import oracle as orc
    lib = group1
    rows, cols = 517, 1031
    hA = oracle.uniform_f32(rows * cols, 3, -1.0, 1.0).reshape(rows, cols)
    hr = oracle.uniform_f32(cols, 4, -1.0, 1.0).reshape(1, cols)
    A, r = lib.to_device(hA), lib.to_device(hr)
    out = lib.empty((rows, cols), np.float32)
    lib.sharded_elementwise(sma.OP_MUL, np.float32, [A.ptr], [cols, 1], [r.ptr], [0, 1], [rows, cols], [out.ptr])
    lib.sharded_synchronize()
    assert np.array_equal(out.numpy(), oracle.binary(orc.MUL, hA, hr))


@pytest.mark.gpu
def test_rank_communicator_one_rank(smhip):
    """The one-process-per-GPU form: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce with a single rank."""
    lib = smhip
    lib.set_device(0)
    uid = lib.comm_unique_id()
    assert len(uid) == 128
    lib.comm_init_rank(1, 0, uid)
    assert lib.comm_info() == (1, 0)
    x = lib.to_device(np.array([1.5, -2.25, 1e300], dtype=np.float64))
    lib.allreduce_sum_async(np.float64, x.ptr, 3)
    assert np.array_equal(x.numpy(), np.array([1.5, -2.25, 1e300]))
    i = lib.to_device(np.array([-7, 2**31 - 1], dtype=np.int32))
    lib.allreduce_sum_async(np.int32, i.ptr, 2)
    assert np.array_equal(i.numpy(), np.array([-7, 2**31 - 1], dtype=np.int32))
    lib.comm_destroy()
    assert lib.comm_info() == (0, -1)
    with pytest.raises(sma.SmhipError, match="smhip_comm_init_rank"):
        lib.allreduce_sum_async(np.float64, x.ptr, 1)


@pytest.mark.gpu
def test_sharded_cpp_surface():
    """sm::set_devices / sm::Sharded<T> (tests/cpp/test_sharded.cpp) on every GPU of the box."""
    from simplemath_amd import build
    build.build_lib()
    exe = build.build_host_programs()["test_sharded"]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 failures" in r.stdout


@pytest.mark.gpu
def test_group_reports_what_rccl_built(group1):
    """smhip_group_info / smhip_rccl_version: RCCL's own answers (ncclCommCount, ncclCommUserRank, ncclCommCuDevice,
    ncclGetVersion) -- what bench.py prints as "rccl" so that an N-GPU line shows the collective saw N ranks."""
    lib = group1
    assert lib.group_info(0) == (1, 0, 0)
    assert lib.rccl_version() >= 20000
    with pytest.raises(sma.SmhipError, match="outside the group"):
        lib.group_info(1)


@pytest.mark.gpu
def test_failed_group_setup_leaves_nothing_behind(smhip, monkeypatch):
    """a failing ncclCommInitAll used to leave the per-device slots allocated and no way to free
    them.  SMHIP_TEST_FAIL_COMM_INIT makes the call fail after the slots exist: they must be back in the pool, no group
    must exist, and a later smhip_set_devices must work."""
    lib = smhip
    lib.set_device(0)
    lib.synchronize()
    in_use_before, _ = lib.pool_stats()
    monkeypatch.setenv("SMHIP_TEST_FAIL_COMM_INIT", "1")
    lib.set_devices(1)  # the variable alone does nothing: the hooks have to be switched on
    lib.set_devices(0)
    lib.c.smhip_enable_test_hooks(1)
    try:
        with pytest.raises(sma.SmhipError, match="ncclCommInitAll"):
            lib.set_devices(1)
    finally:
        lib.c.smhip_enable_test_hooks(0)
    assert lib.get_devices() == 0
    assert lib.pool_stats()[0] == in_use_before
    monkeypatch.delenv("SMHIP_TEST_FAIL_COMM_INIT")
    lib.set_devices(1)
    try:
        x = lib.to_device(np.arange(1000, dtype=np.float32))
        assert lib.sharded_sum(np.float32, [x.ptr], [1000]) == 499500.0
    finally:
        lib.set_devices(0)
    del x
    assert lib.pool_stats()[0] == in_use_before


@pytest.mark.gpu
def test_partly_issued_collective_dissolves_the_group(smhip, monkeypatch):
    """A failing ncclAllReduce inside ncclGroupStart/End must not leave other devices' streams with half a collective:
    the group is closed, its communicators aborted, no group is left; forming a new one works and computes."""
    lib = smhip
    lib.set_device(0)
    lib.set_devices(1)
    x = lib.to_device(np.arange(1000, dtype=np.float32))
    monkeypatch.setenv("SMHIP_TEST_FAIL_ALLREDUCE", "0")
    lib.c.smhip_enable_test_hooks(1)
    try:
        with pytest.raises(sma.SmhipError, match="the device group was dissolved"):
            lib.sharded_sum(np.float32, [x.ptr], [1000])
    finally:
        lib.c.smhip_enable_test_hooks(0)
    assert lib.get_devices() == 0
    monkeypatch.delenv("SMHIP_TEST_FAIL_ALLREDUCE")
    with pytest.raises(sma.SmhipError, match="smhip_set_devices"):
        lib.sharded_sum(np.float32, [x.ptr], [1000])
    lib.set_devices(1)
    try:
        assert lib.sharded_sum(np.float32, [x.ptr], [1000]) == 499500.0
    finally:
        lib.set_devices(0)


@pytest.mark.gpu
def test_copy_peer_is_ordered_on_both_sides(smhip):
    """smhip_copy_peer with source and destination on the one GPU present (the self-peer case): ordered after the kernel
    that produces the source, before the kernel that consumes the destination, and before the source is overwritten."""
    lib = smhip
    lib.set_device(0)
    n = (1 << 22) + 5
    a = lib.uniform_f32(n, 21, -1.0, 1.0)
    src = lib.array_scalar(sma.OP_MUL, a, np.float32(3.0))          # producer, still queued
    dst = lib.empty((n,), np.float32)
    lib.copy_peer(dst.ptr, 0, src.ptr, 0, n * 4)
    lib.array_scalar(sma.OP_MUL, a, np.float32(-1.0), out=src)       # overwrites the source right behind the copy
    twice = lib.array_scalar(sma.OP_ADD, dst, np.float32(1.0))       # consumer of the destination
    ha = a.numpy()
    assert np.array_equal(dst.numpy(), ha * np.float32(3.0))
    assert np.array_equal(twice.numpy(), ha * np.float32(3.0) + np.float32(1.0))
    assert np.array_equal(src.numpy(), ha * np.float32(-1.0))
    with pytest.raises(sma.SmhipError, match="copy_peer"):
        lib.copy_peer(dst.ptr, 0, src.ptr, 99, 16)
