"""Particle sharding across the GPUs of one node: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

The path shards over particles: GP operands are replicated, every rank rolls out P/G particles with its
own noise stream, and ONE collective per CEM iteration assembles the candidate elites: an in-place all-gather of the
ranks' top-k rows [k x (2 + H n_u)] (several problems at once: an all-reduce(sum) over zero-initialised slots).  The
message is a few hundred KB at most -- latency-bound on xGMI -- so nothing larger is ever moved.  After it every rank holds
the same bytes and redundantly picks the global top-k and refits: no second collective, results bit-identical across ranks.
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

    One allocation holds the buffers of ALL iterations (one fill per solve).  A row is `[con, obj, actions...]`; a solve
    has no collective besides its `iterations` exchanges: the per-rank status words ride along with the LAST one.

    ONE problem (E = 1, the BASELINE configs 2-4): iteration `it` owns `[G x (k + 1) x (2 + L)]` rows, rank-major, and the
    exchange is an ALL-GATHER of the ranks' `(k + 1) x (2 + L)` blocks (the ranking kernel writes this rank's block, a
    buffer of its own, directly) -- half the steps and half the bytes of an all-reduce, and the message (56 KB per rank at
    config 2 on 8 GPUs) is latency-bound on xGMI either way.  Row k of a block is padding that keeps the candidate stride
    uniform: `[NaN, NaN, status word, 0...]`, ranked behind everything (NaN sorts last), never an elite while k real rows
    exist; its third cell carries the rank's status word.
    SEVERAL problems at once (E > 1): `[E x G x k x (2 + L)]` slots -- a rank's rows are then strided -- zero-initialised,
    every rank fills only its own, and an all-reduce(sum) assembles them (all-gather-shaped and exact: x + 0 = x for
    finite x and +-inf, NaN stays NaN; -0.0 + 0.0 = +0.0 changes neither the order nor the refit); G trailing cells carry
    the status words.
    """

    def __init__(self, iterations: int, episodes: int, k: int, row_len: int, group, device, dtype=torch.float64):
        self.group = group
        self.world, self.rank = world_and_rank(group)
        self.E, self.k, self.L = episodes, k, row_len
        self.gather = episodes == 1
        W = 2 + row_len
        if self.gather:
            self.rows = k + 1                       # rows a rank hands in (the last one is the padding / status row)
            self.buf = torch.empty((iterations, self.world, self.rows, W), dtype=dtype, device=device)   # gathered
            self.local = torch.zeros((iterations, self.rows, W), dtype=dtype, device=device)             # this rank's blocks
            self.local[:, k, :2] = float('nan')
        else:
            self.rows = k
            self.n_slots = episodes * self.world * k * W
            self.buf = torch.zeros((iterations, self.n_slots + self.world), dtype=dtype, device=device)

    @property
    def candidates(self) -> int:
        """Candidate rows per problem after the exchange."""
        return self.world * self.rows

    def slots(self, it: int) -> Tensor:
        """[E x G x rows x (2 + L)] view of iteration `it`."""
        if self.gather:
            return self.buf[it].unsqueeze(0)
        return self.buf[it, :self.n_slots].view(self.E, self.world, self.k, 2 + self.L)

    def local_slot(self, it: int) -> Tensor:
        """This rank's [E x k x (2 + L)] slot (contiguous when E == 1: the rank kernel then writes straight into it)."""
        if self.gather:
            return self.local[it, :self.k].unsqueeze(0)
        return self.slots(it)[:, self.rank]

    def exchange(self, it: int, status: Optional[Tensor] = None):
        """The ONE collective of iteration `it`.  Returns (candidates [E x G*rows x (2 + L)], status words int32 [G] or
        None): with `status` (this rank's int32 [1] word, passed on the last iteration) every rank learns the words of
        all ranks, so that all ranks raise, or not, together."""
        if self.gather:
            if status is not None:
                self.local[it, self.k, 2] = status[0]
            dist.all_gather_into_tensor(self.buf[it].view(-1), self.local[it].view(-1), group=self.group)
            words = self.buf[it, :, self.k, 2].to(torch.int32) if status is not None else None
            return self.buf[it].view(1, self.world * self.rows, 2 + self.L), words
        if status is not None:
            self.buf[it, self.n_slots + self.rank] = status[0]
        dist.all_reduce(self.buf[it], op=dist.ReduceOp.SUM, group=self.group)
        words = self.buf[it, self.n_slots:].to(torch.int32) if status is not None else None
        return self.slots(it).view(self.E, self.world * self.k, 2 + self.L), words
