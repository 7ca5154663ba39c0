"""GPU, 2 processes sharing the one card, gloo group: the sharded solve end to end (local rollout on the HIP path, local
elite rows, one collective, global ranking on every rank) equals the oracle's solve over the union of the particles,
bit-identically on both ranks.  The driver's multi-GPU run uses the same code with backend nccl = RCCL; NCCL wants one
device per rank, so on this single-GPU box the RCCL calls themselves are exercised with a group of ONE rank
(test_rccl_single_rank_exchange_and_sharded_path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from oracle import cem as ocem
        from oracle.gp import ExactGP
        from safe_exploration_amd import problems
        from safe_exploration_amd.cem_mpc import FusedCemMpc
        dev = torch.device('cuda:0')
        spec = problems.pendulum(n_train=90, seed=4, obj_mode=1)
        ssm, env = problems.build(spec, dev)
        P, H, k, iters = 192, 5, 16, 3            # global particle count, sharded 96 + 96
        rng = np.random.default_rng(33)
        noise = rng.normal(size=(iters, P, H, 1))   # the global draws, identical on both ranks
        x0 = np.array([0.015, -0.02])
        mpc = FusedCemMpc(ssm, env, H, P, k, iters, device=dev, init_std=0.2, process_group=dist.group.WORLD)
        lo, hi = rank * P // WORLD, (rank + 1) * P // WORLD
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        best, ok, _, status = mpc.solve(t(x0[None]), noise=t(noise[:, None, lo:hi]))
        torch.cuda.synchronize()
        gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
        ref, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0, noise, k, init_std=np.full((H, 1), 0.2))
        assert tuple(status.shape) == (WORLD,) and not bool(status.any()) and ref is not None and int(ok[0]) == 1
        np.testing.assert_allclose(best[0].cpu().numpy(), ref, rtol=0, atol=1e-9)
        # two problems at once through the same sharded path (the slot copy of the E > 1 branch)
        x02 = np.array([[0.015, -0.02], [-0.03, 0.04]])
        noise2 = rng.normal(size=(iters, 2, P, H, 1))
        best2, ok2, _, status2 = mpc.solve(t(x02), noise=t(noise2[:, :, lo:hi]))
        assert not bool(status2.any())
        for e in range(2):
            ref_e, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x02[e], noise2[:, e], k,
                                      init_std=np.full((H, 1), 0.2))
            assert (ref_e is not None) == bool(ok2[e])
            if ref_e is not None:
                np.testing.assert_allclose(best2[e].cpu().numpy(), ref_e, rtol=0, atol=1e-9)
        # uneven shards (193 = 97 + 96): every rank still hands in k rows, the global top-k misses nothing
        from safe_exploration_amd import distributed
        P3 = 193
        noise3 = rng.normal(size=(iters, P3, H, 1))
        mpc3 = FusedCemMpc(ssm, env, H, P3, k, iters, device=dev, init_std=0.2, process_group=dist.group.WORLD)
        cnt, off = distributed.shard_particles(P3, WORLD, rank)
        best3, ok3, _, status3 = mpc3.solve(t(x0[None]), noise=t(noise3[:, None, off:off + cnt]))
        ref3, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0, noise3, k, init_std=np.full((H, 1), 0.2))
        assert not bool(status3.any()) and (ref3 is not None) == bool(ok3[0])
        if ref3 is not None:
            np.testing.assert_allclose(best3[0].cpu().numpy(), ref3, rtol=0, atol=1e-9)
        out[rank] = best[0].cpu().numpy().tobytes() + best2.cpu().numpy().tobytes() + best3.cpu().numpy().tobytes()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_solve_two_ranks():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
        assert len(out) == WORLD and out[0] == out[1]        # every rank holds the same bytes


def _rccl_worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from oracle import cem as ocem
        from oracle.gp import ExactGP
        from safe_exploration_amd import distributed, problems
        from safe_exploration_amd.cem_mpc import FusedCemMpc
        dev = torch.device('cuda:0')
        assert dist.get_backend() == 'nccl'
        # the collectives as EliteExchange issues them: all-gather of the (k + 1)-row blocks (one problem), all-reduce over
        # the zero-padded slots (two problems), status words riding along
        rng = np.random.default_rng(2)
        for E in (1, 2):
            k, L = 7, 5
            xch = distributed.EliteExchange(2, E, k, L, dist.group.WORLD, dev)
            rows = rng.normal(size=(E, k, 2 + L))
            xch.local_slot(1).copy_(torch.tensor(rows, device=dev))
            cand, words = xch.exchange(1, torch.tensor([6], dtype=torch.int32, device=dev))
            torch.cuda.synchronize()
            assert words.tolist() == [6] and tuple(cand.shape) == (E, xch.candidates, 2 + L)
            np.testing.assert_array_equal(cand[:, :k].cpu().numpy(), rows)
            if E == 1:
                assert bool(torch.isnan(cand[0, k, :2]).all())          # the padding / status row
        # the solver's sharded code path (local ranking -> RCCL collective -> global ranking -> prologue refit) on one rank
        spec = problems.pendulum(n_train=90, seed=4, obj_mode=1)
        ssm, env = problems.build(spec, dev)
        P, H, k, iters = 192, 5, 16, 3
        noise = rng.normal(size=(iters, P, H, 1))
        x0 = np.array([0.015, -0.02])
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        mpc = FusedCemMpc(ssm, env, H, P, k, iters, device=dev, init_std=0.2, process_group=dist.group.WORLD,
                          force_exchange=True)
        best, ok, _, status = mpc.solve(t(x0[None]), noise=t(noise[:, None]))
        torch.cuda.synchronize()
        gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
        ref, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0, noise, k, init_std=np.full((H, 1), 0.2))
        assert tuple(status.shape) == (1,) and not bool(status.any()) and ref is not None and int(ok[0]) == 1
        np.testing.assert_allclose(best[0].cpu().numpy(), ref, rtol=0, atol=1e-9)
        out[0] = 'ok'
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_single_rank_exchange_and_sharded_path():
    """backend 'nccl' IS RCCL on ROCm: the exchange's collectives and the solver's sharded path over it, with a group of one
    rank (all this box can host).  What it cannot show is the time the collective takes across GPUs."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
        assert out.get(0) == 'ok'


@pytest.mark.timeout(400)
def test_sharded_solve_fuzz_small():
    """A short run of tools/dist_fuzz.py: random shapes (uneven shards, elites up to the smaller share, one or two problems)
    through the two-rank sharded solve, against the oracle over the union of the particles, bit-identical on both ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dist_fuzz.py'), '12', '5'], capture_output=True, text=True,
                       timeout=380)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
