"""Particle sharding across the GPUs of one node: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

The path shards over particles: GP operands are replicated, every rank rolls out P/G particles with its
own noise stream, and ONE collective per CEM iteration assembles the candidate elites: an all-reduce(sum) over a
zero-initialised [G x k x (2 + H n_u)] buffer in which each rank fills only its own slot.  The message is a few
hundred KB at most -- latency-bound on xGMI -- so nothing larger is ever reduced.  After it every rank holds the same
bytes and redundantly picks the global top-k and refits: no second collective, results bit-identical across ranks.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def world_and_rank(group=None) -> Tuple[int, int]:
    if group is None:
        return 1, 0   # sharding is opt-in: pass the group (e.g. dist.group.WORLD) explicitly
    return dist.get_world_size(group), dist.get_rank(group)


def shard_particles(total: int, world: int, rank: int) -> Tuple[int, int]:
    """(count, offset) of this rank's particles; the first `total % world` ranks take one extra."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return count, offset


def rank_seed(seed: int, rank: int) -> int:
    """Per-rank noise stream (cfg 3 seeds 1000 + rank)."""
    return int(seed) + 1000 * int(rank)


def exchange_elite_rows(rows: Tensor, group=None) -> Tensor:
    """rows [E x k x W] (this rank's local elites, best first) -> [E x G*k x W], rank-major, identical on every rank.

    One all-reduce(sum) over zero-padded slots.  Adding zeros is exact in IEEE arithmetic for finite values and keeps
    +-inf; a NaN cost stays NaN.  (-0.0 + 0.0 = +0.0 does not change any ordering or refit.)
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    E, k, W = rows.shape
    buf = torch.zeros((E, world, k, W), dtype=rows.dtype, device=rows.device)
    buf[:, rank] = rows
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf.view(E, world * k, W)


def all_reduce_slots(buf: Tensor, group=None) -> Tensor:
    """buf [E x G x k x W], zero everywhere but in this rank's slot [:, rank] -> the same buffer after ONE
    all-reduce(sum), viewed [E x G*k x W] (what exchange_elite_rows returns, without its allocation and copy)."""
    E, world, k, W = buf.shape
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf.view(E, world * k, W)


def all_reduce_sum_(buf: Tensor, group=None) -> None:
    """In-place all-reduce(sum) of a buffer in which every rank has filled only its own cells (zeros elsewhere):
    all-gather-shaped, and adding zeros is exact."""
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)


def all_reduce_max_(t: Tensor, group=None) -> None:
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
