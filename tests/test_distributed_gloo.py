"""CPU, 2 processes, gloo: the N > 1 data path of the solver -- shard the particles, keep local elites, ONE all-reduce over
zero-padded slots, every rank ranks the same candidates.  The kernels' part (local ranking) is played by the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cem as ocem
from safe_exploration_amd import distributed

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, port, P, k, L, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        rng = np.random.default_rng(7)                 # the same global population on both ranks
        con = rng.choice([0., 0., 3., 10.], size=P)
        obj = rng.normal(size=P)
        act = rng.normal(size=(P, L))
        count, off = distributed.shard_particles(P, WORLD, rank)
        idx = ocem.rank(con[off:off + count], obj[off:off + count], k) + off       # local elites (the kernel's job)
        rows = np.concatenate((con[idx, None], obj[idx, None], act[idx]), axis=1)   # [k x (2 + L)]
        cand = distributed.exchange_elite_rows(torch.tensor(rows[None]), dist.group.WORLD)[0].numpy()
        assert cand.shape == (WORLD * k, 2 + L)
        np.testing.assert_array_equal(cand[rank * k:(rank + 1) * k], rows)         # own slot untouched
        # the form the solver uses: the rows are written straight into this rank's slot of a zeroed buffer
        buf = torch.zeros((1, WORLD, k, 2 + L), dtype=torch.float64)
        buf[0, rank] = torch.tensor(rows)
        cand2 = distributed.all_reduce_slots(buf, dist.group.WORLD)
        assert cand2.data_ptr() == buf.data_ptr()                                   # in place, no copy
        np.testing.assert_array_equal(cand2[0].numpy(), cand)
        # global selection from the candidates == selection from the whole population
        sel = ocem.rank(cand[:, 0], cand[:, 1], k)
        want = ocem.rank(con, obj, k)
        np.testing.assert_array_equal(cand[sel][:, 2:], act[want])
        mean, std = ocem.refit(cand[sel][:, 2:].reshape(k, L, 1))
        st = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
        distributed.all_reduce_max_(st, dist.group.WORLD)
        assert int(st) == 1
        out[rank] = np.concatenate((mean.ravel(), std.ravel())).tobytes()           # must be bit-identical on all ranks
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_elite_exchange_world2():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(port, 1001, 16, 5, out), nprocs=WORLD, join=True)
        assert len(out) == WORLD and out[0] == out[1]
