"""Particle sharding across the GPUs of one node: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

The path shards over particles: GP operands are replicated, every rank rolls out P/G particles with its
own noise stream, and ONE collective per CEM iteration assembles the candidate elites: an in-place all-gather of the
ranks' top-k rows [k x (2 + H n_u)] (several problems at once: an all-reduce(sum) over zero-initialised slots).  The
message is a few hundred KB at most -- latency-bound on xGMI -- so nothing larger is ever moved.  After it every rank holds
the same bytes and redundantly picks the global top-k and refits: no second collective, results bit-identical across ranks.
"""
import ctypes
import os
import weakref
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_byte * 128)]   # (c_byte: a c_char array reads back truncated at the first NUL)


class RcclComm:
    """An RCCL communicator of our own, driven through ctypes from the librccl.so torch ships, so that the ONE collective of
    a CEM iteration is enqueued on the COMPUTE stream like any kernel: rollout -> local ranking -> all-gather -> global
    ranking run back to back on one stream.  torch's NCCL process group runs collectives on a stream of its own and brackets
    each with two cross-stream event waits and a Python `Work` object: with ONE rank (no inter-GPU latency at all) that
    plumbing was a third of the sharded path's fixed cost of 15 us per iteration.

    Created collectively (every rank of `group` constructs it): rank 0's ncclUniqueId travels over the existing process
    group (any backend), then ncclCommInitRank.  One device per rank, as RCCL requires.  SX_RCCL_DIRECT=0 turns the direct
    path off (FusedCemMpc then uses torch.distributed's collectives)."""
    FLOAT64, SUM = 8, 0     # ncclFloat64, ncclSum (nccl.h)

    _lib = None

    @classmethod
    def library(cls):
        if cls._lib is None:
            path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            lib = ctypes.CDLL(path)     # (already mapped by torch: the same library instance)
            lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
            lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p]
            lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            lib.ncclGetErrorString.argtypes = [ctypes.c_int]
            lib.ncclGetErrorString.restype = ctypes.c_char_p
            for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllGather, lib.ncclAllReduce, lib.ncclCommDestroy):
                f.restype = ctypes.c_int
            cls._lib = lib
        return cls._lib

    @classmethod
    def wanted(cls, group, device) -> bool:
        """Direct RCCL where the group's backend is nccl (one device per rank) and it is not switched off."""
        if group is None or os.environ.get('SX_RCCL_DIRECT', '1') == '0' or torch.device(device).type != 'cuda':
            return False
        try:
            return 'nccl' in str(dist.get_backend(group))
        except Exception:   # noqa: BLE001
            return False

    def __init__(self, group, device):
        lib = self.library()
        self.device = torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        uid = _NcclUniqueId()
        if self.rank == 0:
            self._check(lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        payload = [ctypes.string_at(ctypes.byref(uid), 128)] if self.rank == 0 else [None]
        if self.world > 1:
            dist.broadcast_object_list(payload, src=dist.get_global_rank(group, 0), group=group)
        assert len(payload[0]) == 128
        ctypes.memmove(ctypes.byref(uid), payload[0], 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), 'ncclCommInitRank')
        self._finalizer = weakref.finalize(self, lib.ncclCommDestroy, self._comm)

    def _check(self, code: int, what: str) -> None:
        if code != 0:
            raise RuntimeError(f'{what} failed: {self.library().ncclGetErrorString(code).decode()}')

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_gather(self, send: Tensor, recv: Tensor) -> None:
        """recv [world x send.numel()] <- every rank's send, enqueued on the current stream (float64, contiguous)."""
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == self.world * send.numel()
        self._check(self.library().ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self.FLOAT64, self._comm,
                                                 self._stream()), 'ncclAllGather')

    def all_reduce_sum_(self, buf: Tensor) -> None:
        assert buf.is_contiguous()
        self._check(self.library().ncclAllReduce(buf.data_ptr(), buf.data_ptr(), buf.numel(), self.FLOAT64, self.SUM,
                                                 self._comm, self._stream()), 'ncclAllReduce')


def make_comm(group, device) -> Optional[RcclComm]:
    """The direct RCCL communicator for `group`, or None (then the exchange goes through torch.distributed).  Collective: if
    ANY rank fails to create its communicator -- a second rank on the same device, no usable bootstrap interface -- every
    rank drops to torch's path together (one all-reduce of a flag over the existing group) and rank 0 says so once."""
    if not RcclComm.wanted(group, device):
        return None
    comm, err = None, ''
    try:
        comm = RcclComm(group, device)
    except Exception as exc:   # noqa: BLE001
        err = str(exc)
    flag = torch.tensor([0 if comm is not None else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()) != 0:
        if dist.get_rank(group) == 0:
            import sys
            print(f'safe_exploration_amd: direct RCCL communicator not available ({err or "another rank failed"}); '
                  f'the elite exchange uses torch.distributed', file=sys.stderr)
        return None
    return comm


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

    def __init__(self, iterations: int, episodes: int, k: int, row_len: int, group, device, dtype=torch.float64, comm=None):
        self.group = group
        self.comm = comm     # an RcclComm: the collective goes straight onto the compute stream
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
            if self.comm is not None:
                self.comm.all_gather(self.local[it].view(-1), self.buf[it].view(-1))
            else:
                dist.all_gather_into_tensor(self.buf[it].view(-1), self.local[it].view(-1), group=self.group)
            words = self.buf[it, :, self.k, 2].to(torch.int32) if status is not None else None
            return self.buf[it].view(1, self.world * self.rows, 2 + self.L), words
        if status is not None:
            self.buf[it, self.n_slots + self.rank] = status[0]
        if self.comm is not None:
            self.comm.all_reduce_sum_(self.buf[it])
        else:
            dist.all_reduce(self.buf[it], op=dist.ReduceOp.SUM, group=self.group)
        words = self.buf[it, self.n_slots:].to(torch.int32) if status is not None else None
        return self.slots(it).view(self.E, self.world * self.k, 2 + self.L), words
