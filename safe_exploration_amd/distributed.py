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


class EliteExchange:
    """The per-iteration exchange of one sharded solve: what `FusedCemMpc.solve` runs between its local
    and its global ranking launch, and what tests/test_distributed_gloo.py drives on the CPU.

    One allocation holds the zero-initialised buffers of ALL iterations (one memset per solve).  Iteration `it` owns
    `[E x G x k x (2 + L)]` candidate slots -- a row is `[con, obj, actions...]` -- plus G trailing cells in which the
    per-rank status words ride along with the LAST exchange, so a solve has no collective besides its `iterations`
    all-reduces.  Every rank fills only its own slot; the all-reduce(sum) over the zero padding is all-gather-shaped and
    exact (x + 0 = x for finite x and +-inf, NaN stays NaN; -0.0 + 0.0 = +0.0 changes neither the order nor the refit).
    """

    def __init__(self, iterations: int, episodes: int, k: int, row_len: int, group, device, dtype=torch.float64):
        self.group = group
        self.world, self.rank = world_and_rank(group)
        self.E, self.k, self.L = episodes, k, row_len
        self.n_slots = episodes * self.world * k * (2 + row_len)
        self.buf = torch.zeros((iterations, self.n_slots + self.world), dtype=dtype, device=device)

    def slots(self, it: int) -> Tensor:
        """[E x G x k x (2 + L)] view of iteration `it`."""
        return self.buf[it, :self.n_slots].view(self.E, self.world, self.k, 2 + self.L)

    def local_slot(self, it: int) -> Tensor:
        """This rank's [E x k x (2 + L)] slot (contiguous when E == 1: the rank kernel then writes straight into it)."""
        return self.slots(it)[:, self.rank]

    def exchange(self, it: int, status: Optional[Tensor] = None):
        """The ONE collective of iteration `it`.  Returns (candidates [E x G*k x (2 + L)], status words int32 [G] or
        None): with `status` (this rank's int32 [1] word, passed on the last iteration) every rank learns the words of
        all ranks, so that all ranks raise, or not, together."""
        if status is not None:
            self.buf[it, self.n_slots + self.rank] = status[0]
        dist.all_reduce(self.buf[it], op=dist.ReduceOp.SUM, group=self.group)
        words = self.buf[it, self.n_slots:].to(torch.int32) if status is not None else None
        return self.slots(it).view(self.E, self.world * self.k, 2 + self.L), words
