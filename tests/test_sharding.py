"""The N > 1 path without GPUs: outer-dimension shard planning and the one-scalar
all-reduce, over gloo with world_size 2 and 3.  Each rank computes its shard with the
oracle (standing in for the device kernels, which the gpu-marked tests cover), the
shards are stitched / all-reduced, and the whole is compared with the oracle on the
unsharded problem."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from simplemath_amd import sharding  # noqa: E402


def test_split_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for world in (1, 2, 3, 8):
            blocks = [sharding.split_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_shard_outer_offsets_and_replication():
    # config 3 shape: A (4096 x 4096) dense, r (1 x 4096) broadcast along dim 0 -> replicated
    sh = sharding.shard_outer((4096, 4096), (4096, 1), (0, 1), world=8, rank=3)
    assert sh.shape == (512, 4096) and sh.start == 1536
    assert sh.offset_a == 1536 * 4096 and sh.offset_b == 0 and sh.offset_out == 1536 * 4096
    assert sh.replicated_b and not sh.replicated_a
    # 1-D config 5: contiguous ranges
    sh = sharding.shard_outer((1 << 31,), (1,), (1,), world=8, rank=7)
    assert sh.shape == (1 << 28,) and sh.offset_a == 7 << 28 and sh.size == 1 << 28
    # dim 0 shorter than the world: trailing ranks are empty
    assert sharding.shard_outer((2, 5), (5, 1), (5, 1), world=4, rank=3).size == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from tests.golden import gen
        o = orc.Oracle()
        fails = []

        # (1) broadcast multiply, config-3 shape in miniature: shards stitch to the unsharded result
        rows, cols = 37, 64
        A = gen.gen(np.float32, rows * cols, 1, "uniform").reshape(rows, cols)
        r = gen.gen(np.float32, cols, 2, "uniform").reshape(1, cols)
        shape, sa, sb, _ = o.broadcast(A.shape, [cols, 1], r.shape, [cols, 1])
        sh = sharding.shard_outer(shape, sa, sb, world, rank)
        # the planner the product uses (libsmhip's smhip_shard_outer: host-only code, so it runs here without a GPU) must
        # cut the same block; the shard below is taken from ITS answer
        import simplemath_amd as sma
        c_shape, c_off_a, c_off_b, c_off_out, c_rep_a, c_rep_b = sma.load().shard_outer(shape, sa, sb, world, rank)
        if (c_shape, c_off_a, c_off_b, c_off_out, c_rep_a, c_rep_b) != (sh.shape, sh.offset_a, sh.offset_b, sh.offset_out, sh.replicated_a, sh.replicated_b):
            fails.append("C planner and Python planner disagree")
        a_sh = A.reshape(-1)[c_off_a:]
        b_sh = r.reshape(-1)[c_off_b:]
        part = o.elementwise(orc.MUL, a_sh, sa, b_sh, sb, list(c_shape)) if sh.size else np.empty(0, np.float32)
        np.save(os.path.join(tmpdir, f"part{rank}.npy"), part)
        dist.barrier()
        if rank == 0:
            whole = np.concatenate([np.load(os.path.join(tmpdir, f"part{k}.npy")) for k in range(world)])
            if not np.array_equal(whole, o.binary(orc.MUL, A, r).reshape(-1)):
                fails.append("stitched broadcast multiply differs")

        # (2) config 5 in miniature: fused add + sum per shard, ONE all-reduce of the fp64 scalar
        n = 100003
        first, count = sma.load().split_range(n, world, rank)  # smhip_split_range
        if (first, count) != sharding.split_range(n, world, rank):
            fails.append("C and Python split_range disagree")
        a = o.uniform_f32(count, 6, 0.0, 1.0, first=first)   # each rank generates ITS slice of the global stream
        b = o.uniform_f32(count, 7, 0.0, 1.0, first=first)
        _, partial = o.contiguous_sum(orc.ADD, a, b)
        total = sharding.allreduce_scalar(partial, "f64", dist)
        fa, fb = o.uniform_f32(n, 6, 0.0, 1.0), o.uniform_f32(n, 7, 0.0, 1.0)
        _, want = o.contiguous_sum(orc.ADD, fa, fb)
        if abs(total - want) > 1e-9 * abs(want):
            fails.append(f"all-reduced sum {total} vs {want}")

        # (3) wrapping int32 dot: per-rank partials add modulo 2^32 to the reference's value
        ia = gen.gen(np.int32, n, 3, "wide")
        ib = gen.gen(np.int32, n, 4, "wide")
        p = int(o.dot(ia[first:first + count], ib[first:first + count])) if count else 0
        got = sharding.allreduce_scalar(p, "i32", dist)
        if got != int(o.dot(ia, ib)):
            fails.append(f"int32 dot {got} vs {int(o.dot(ia, ib))}")

        with open(os.path.join(tmpdir, f"result{rank}.txt"), "w") as f:
            f.write("\n".join(fails))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_path_over_gloo(world, tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"result{r}.txt").read_text() == "", f"rank {r}"
