"""Outer-dimension sharding of the hot path across the GPUs of one node.

One process per GPU.  Elementwise outputs are independent, so each rank runs the
single-GPU kernels on its block of the RESULT's outermost dimension and there is no
data-path collective.  Only the whole-array reductions (sum, dot, fused op+sum) have an
exchange step: ONE all-reduce of an 8-byte scalar per rank (fp64 for float types,
int64 for the wrapping integer dot), latency-bound, over RCCL/xGMI on GPUs ("nccl"
backend) or gloo in the CPU tests.

Nothing here touches a GPU: it is index arithmetic plus torch.distributed calls, which
is why the world_size-2 gloo tests can cover it (tests/test_sharding.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence


@dataclass(frozen=True)
class Shard:
    """Rank `rank`'s block of a broadcasted a-op-b problem."""
    rank: int
    world: int
    shape: tuple          # this rank's result shape (dim 0 cut down)
    start: int            # first index of the block along result dim 0
    offset_a: int         # element offsets into the operands / the dense result
    offset_b: int
    offset_out: int
    replicated_a: bool    # operand is broadcast along dim 0: every rank reads all of it
    replicated_b: bool

    @property
    def size(self):
        n = 1
        for d in self.shape:
            n *= d
        return n


def split_range(n: int, world: int, rank: int):
    """Near-equal contiguous blocks: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def shard_outer(shape: Sequence[int], strides_a: Sequence[int], strides_b: Sequence[int], world: int, rank: int) -> Shard:
    """Cut the result's outermost dimension into `world` blocks.

    `shape`, `strides_*` are the broadcast result shape and the operands' broadcast strides
    (elements), i.e. what sm::broadcast / smhip_broadcast return.  If dim 0 is shorter than
    `world`, trailing ranks get empty shards (callers may first flatten leading dims).
    """
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    shape = tuple(int(s) for s in shape)
    start, count = split_range(shape[0], world, rank)
    inner = 1
    for d in shape[1:]:
        inner *= d
    return Shard(rank=rank, world=world, shape=(count,) + shape[1:], start=start,
                 offset_a=start * int(strides_a[0]), offset_b=start * int(strides_b[0]), offset_out=start * inner,
                 replicated_a=int(strides_a[0]) == 0 and shape[0] > 1, replicated_b=int(strides_b[0]) == 0 and shape[0] > 1)


def allreduce_scalar(partial, kind: str, dist, device=None):
    """Combine per-rank partial reductions.  kind: "f64" (sum / float dot / fused op+sum),
    "i32" / "i64" (wrapping integer dot: partials add modulo 2^32 / 2^64, so any order and any
    sharding reproduces the single-GPU -- and the reference's -- result bit for bit)."""
    import torch
    if kind == "f64":
        t = torch.tensor([float(partial)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0])
    t = torch.tensor([int(partial)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # int64 addition wraps modulo 2^64
    v = int(t[0])
    if kind == "i32":
        v &= 0xFFFFFFFF
        return v - (1 << 32) if v & 0x80000000 else v
    if kind == "i64":
        return v
    raise ValueError(kind)
