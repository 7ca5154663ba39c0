"""Randomised check of the SHARDED solve: two ranks share the one card under gloo (NCCL wants a device per rank), random
problem shapes (uneven shards, elites up to the smaller share, one or two problems at once); every rank's result must be
the oracle's solve over the union of the particles, and identical bytes on both ranks.  python tools/dist_fuzz.py [cases] [seed]"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORLD = 2


def worker(rank, port, cases, seed, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from oracle import cem as ocem
        from oracle.gp import ExactGP
        from safe_exploration_amd import distributed, problems
        from safe_exploration_amd.cem_mpc import FusedCemMpc
        dev = torch.device('cuda:0')
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        rng = np.random.default_rng(seed)          # the same stream on both ranks
        bad, digest = 0, b''
        for case in range(cases):
            spec = problems.pendulum(n_train=int(rng.integers(20, 120)), seed=int(rng.integers(0, 100)), obj_mode=int(rng.integers(0, 2)))
            ssm, env = problems.build(spec, dev)
            gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
            E = int(rng.choice([1, 1, 2]))
            P = int(rng.integers(4, 300))
            H, iters = int(rng.integers(1, 7)), int(rng.integers(1, 4))
            k = int(rng.integers(1, P // WORLD + 1))
            noise = rng.normal(size=(iters, E, P, H, 1))
            x0 = rng.normal(0, 0.03, size=(E, 2))
            cnt, off = distributed.shard_particles(P, WORLD, rank)
            mpc = FusedCemMpc(ssm, env, H, P, k, iters, device=dev, init_std=0.2, process_group=dist.group.WORLD)
            best, ok, _, status = mpc.solve(t(x0), noise=t(noise[:, :, off:off + cnt]))
            torch.cuda.synchronize()
            good = tuple(status.shape) == (WORLD,) and not bool(status.any())
            for e in range(E):
                ref, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0[e], noise[:, e], k, init_std=np.full((H, 1), 0.2))
                good = good and (ref is not None) == bool(ok[e])
                if good and ref is not None:
                    good = float(np.abs(best[e].cpu().numpy() - ref).max()) < 1e-8
            if not good:
                bad += 1
                print(f'rank {rank} MISMATCH case {case}: E={E} P={P} H={H} k={k} iters={iters}', flush=True)
            digest += best.cpu().numpy().tobytes() + ok.cpu().numpy().tobytes()
        out[rank] = (bad, digest)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(worker, args=(port, cases, seed, out), nprocs=WORLD, join=True)
        bad = out[0][0] + out[1][0]
        same = out[0][1] == out[1][1]
        print(f'{cases} cases on {WORLD} ranks: {bad} mismatches, ranks bit-identical: {same}')
        sys.exit(0 if bad == 0 and same else 1)
