"""CPU, 2 processes, gloo: the N > 1 data path of the solver.  `distributed.EliteExchange` is the code
`FusedCemMpc.solve` runs between its local and its global ranking launch (particle sharding, ONE collective per
iteration -- an all-gather of the ranks' elite blocks for one problem, an all-reduce over zero-padded slots for several --
status words riding along with the last one); here the two launches around it are played by the oracle's ranking, so
the whole exchange runs without a GPU."""
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


def _worker(rank, port, P, k, L, E, iters, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        xch = distributed.EliteExchange(iters, E, k, L, dist.group.WORLD, 'cpu')
        nrows = k + 1 if E == 1 else k              # one problem: a padding / status row behind every rank's k rows
        assert (xch.world, xch.rank, xch.rows) == (WORLD, rank, nrows)
        assert tuple(xch.slots(0).shape) == (E, WORLD, nrows, 2 + L) and xch.candidates == WORLD * nrows
        count, off = distributed.shard_particles(P, WORLD, rank)   # uneven when P is odd: the first rank takes one more
        digest = b''
        for it in range(iters):
            rng = np.random.default_rng(7 + it)                    # the same global population on both ranks
            con = rng.choice([0., 0., 3., 10.], size=(E, P))
            obj = rng.normal(size=(E, P))
            act = rng.normal(size=(E, P, L))
            for e in range(E):                                     # local elites (on the GPU: cem_rank_kernel's job)
                idx = ocem.rank(con[e, off:off + count], obj[e, off:off + count], k) + off
                rows = np.concatenate((con[e, idx, None], obj[e, idx, None], act[e, idx]), axis=1)   # [k x (2 + L)]
                xch.local_slot(it)[e] = torch.tensor(rows)
            last = it == iters - 1
            status = torch.tensor([4 if rank == 1 else 0], dtype=torch.int32)
            before = xch.local_slot(it).clone()
            cand, words = xch.exchange(it, status if last else None)
            assert cand.data_ptr() == xch.buf[it].data_ptr()                       # the collective's own buffer, no copy
            assert tuple(cand.shape) == (E, WORLD * nrows, 2 + L)
            np.testing.assert_array_equal(xch.local_slot(it).numpy(), before.numpy())   # own slot untouched
            np.testing.assert_array_equal(xch.slots(it)[:, rank, :k].numpy(), before.numpy())   # and where it belongs
            if last:
                assert words.dtype == torch.int32 and words.tolist() == [0, 4]     # every rank sees every rank's word
            else:
                assert words is None
            for e in range(E):
                c = cand[e].numpy()
                # global selection from the G k candidates == selection from the whole population
                sel = ocem.rank(c[:, 0], c[:, 1], k)       # (padding rows are [NaN, NaN, ...]: ranked last)
                assert not np.isnan(c[sel][:, :2]).any()
                want = ocem.rank(con[e], obj[e], k)
                np.testing.assert_array_equal(c[sel][:, 2:], act[e, want])
                mean, std = ocem.refit(c[sel][:, 2:].reshape(k, L, 1))
                digest += np.concatenate((mean.ravel(), std.ravel())).tobytes()
        out[rank] = digest                                                          # must be bit-identical on all ranks
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize('P,k,L,E,iters', [(1000, 16, 5, 1, 2), (193, 24, 3, 2, 3)])
def test_elite_exchange_world2(P, k, L, E, iters):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(port, P, k, L, E, iters, out), nprocs=WORLD, join=True)
        assert len(out) == WORLD and out[0] == out[1]


def test_constructor_refuses_what_the_exchange_cannot_serve():
    """More elites than the smallest per-GPU share would make a rank hand in fewer than k rows: refused up
    front, with the kernel's limits (no GPU needed: the checks run before anything touches the device)."""
    from safe_exploration_amd import cem_mpc

    class _Ssm:
        num_states, num_actions = 2, 1

    class _Group:
        pass

    orig = distributed.world_and_rank
    distributed.world_and_rank = lambda g: (2, 0) if g is not None else (1, 0)
    try:
        with pytest.raises(ValueError, match='smallest per-GPU share'):
            cem_mpc.FusedCemMpc(_Ssm(), None, 5, 5, 3, 2, device='cpu', process_group=_Group())
        cem_mpc.FusedCemMpc(_Ssm(), None, 5, 40000, 10, 2, device='cpu')          # ranks in 4 chunks of 10 000
        assert cem_mpc.rank_chunks(16384) == 1 and cem_mpc.rank_chunks(65536) == 4 and cem_mpc.rank_chunks(40000) == 4
        with pytest.raises(ValueError, match='too large'):
            cem_mpc.FusedCemMpc(_Ssm(), None, 5, 262144, 2048, 2, device='cpu')      # 16 chunks x 2048 rows > 16 384
        with pytest.raises(ValueError, match='limit'):
            cem_mpc.FusedCemMpc(_Ssm(), None, 5, 16384, 4096, 2, device='cpu')
        with pytest.raises(ValueError, match='warm_start'):
            cem_mpc.FusedCemMpc(_Ssm(), None, 5, 64, 8, 2, device='cpu', warm_start='lqr')
    finally:
        distributed.world_and_rank = orig
