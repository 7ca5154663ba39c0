#!/usr/bin/env python
"""Headline benchmark: CEM particle-step evaluations/s of the fused safe-MPC solve (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One STEP = one complete MPC solve (`get_action`'s optimiser call): `iters` CEM iterations, each = draw the action noise,
roll all particles out for H steps (GP predict + one-step reachability + costs, one fused launch), rank, refit.
Weak scaling: every GPU holds P particles (N GPUs optimise over N*P particles) and the ranks exchange their elite rows
with ONE all-reduce per CEM iteration.  Inputs are synthetic (seeded) and resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic flops per particle-step  F = n_s [2N^2 + 2N + 2N + 2ND + 3ND + N]
def algorithmic_flops_per_particle_step(n_s, n_train, d_in):
    return n_s * (2 * n_train ** 2 + 2 * n_train + 2 * n_train + 2 * n_train * d_in + 3 * n_train * d_in + n_train)


F64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, datasheet; DESIGN.md "Roofline"


def pmc_mfma_instructions(workload_key):
    """v_mfma_f64_16x16x4 instructions per launch of the dominant kernel (SQ_INSTS_MFMA, same committed PMC passes), or
    None: 2048 flop each -- what the kernel EXECUTES, as opposed to the algorithmic count the roofline contract uses."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')) as f:
            d = json.load(f)
        if d.get('workload') == workload_key:
            return d['per_launch_averages']['cem_rollout_kernel<2,1>']['SQ_INSTS_MFMA']
    except Exception:
        pass
    return None


def pmc_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/), or None.
    bench.py cannot collect PMC counters itself; the number is only reported for the workload it was collected on."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')) as f:
            d = json.load(f)
        if d.get('workload') == workload_key:
            return d['hbm_traffic_bytes_per_launch']['cem_rollout_kernel<2,1>']
    except Exception:
        pass
    return None



def cpu_baseline(spec, horizon, particles, elites, budget_s=12.0, max_iters=64):
    """The oracle's C restatement (oracle/csrc, OpenMP over particles; a port of the reference's arithmetic) on the host
    cores, on a bounded sample of the same workload: whole CEM iterations until `budget_s` seconds are spent."""
    import numpy as np
    from oracle import c_oracle
    from oracle import cem as ocem
    from oracle.gp import ExactGP
    from safe_exploration_amd import problems
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    prob = problems.oracle_problem(spec, ocem)
    rng = np.random.default_rng(1)
    mean, std = np.zeros((horizon, spec.n_u)), np.full((horizon, spec.n_u), 0.1)
    x0 = np.array([0.02, -0.03] + [0.0] * (spec.n_s - 2))[:spec.n_s]
    c_oracle.rollout(prob, gp, x0, mean[None] + std[None] * rng.normal(size=(64, horizon, spec.n_u)), want_traj=False)
    done, t0 = 0, time.perf_counter()
    while done < max_iters and (time.perf_counter() - t0) < budget_s:
        acts = mean[None] + std[None] * rng.normal(size=(particles, horizon, spec.n_u))
        res = c_oracle.rollout(prob, gp, x0, acts, want_traj=False)
        idx = ocem.rank(res.con_cost, res.obj_cost, elites)
        mean, std = ocem.refit(acts[idx])
        done += 1
    dt = time.perf_counter() - t0
    return {'value': particles * horizon * done / dt, 'unit': 'particle-steps/s', 'cores': c_oracle.max_threads(),
            'kind': 'port',
            'sample': f'{done} CEM iteration(s) of the same workload ({particles} particles x H={horizon}), C oracle '
                      f'(oracle/csrc, OpenMP, float64), {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--particles', type=int, default=4096, help='per GPU')
    ap.add_argument('--horizon', type=int, default=15)
    ap.add_argument('--n-train', type=int, default=200)
    ap.add_argument('--iters', type=int, default=8, help='CEM iterations per solve (reference default 8)')
    ap.add_argument('--elites', type=int, default=0, help='0 = 10 %% of the per-GPU particle count')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc, fold_status

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: safe_exploration_amd has no CPU path')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    group = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
        group = dist.group.WORLD

    spec = problems.pendulum(n_train=args.n_train, seed=0)
    ssm, env = problems.build(spec, dev)
    P, H, iters = args.particles, args.horizon, args.iters
    total_particles = P * world
    # the elite count does not grow with the GPU count: every rank contributes its local top-k rows, so the per-iteration
    # all-reduce stays at G x k x (2 + H n_u) doubles (445 KB at 8 GPUs) -- latency-bound on xGMI, as SURVEY 8e asks
    elites = min(args.elites or max(1, P // 10), 2048)
    mpc = FusedCemMpc(ssm, env, H, total_particles, elites, iters, device=dev, seed=1, init_std=0.1, process_group=group)
    x0 = torch.tensor([[0.02, -0.03]], dtype=torch.float64, device=dev)

    def barrier():
        if world > 1:
            dist.barrier(group)
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        mpc.solve(x0)
    mpc.rollout_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best, ok, _, status = mpc.solve(x0)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    status_word = fold_status(status.cpu())
    rollout_ms = [a.elapsed_time(b) for a, b in mpc.rollout_events]
    mpc.rollout_events = None
    avg_rollout_s = sum(rollout_ms) / len(rollout_ms) * 1e-3

    if rank == 0:
        particle_steps = total_particles * H * iters * args.steps
        flops_unit = algorithmic_flops_per_particle_step(spec.n_s, args.n_train, spec.n_s + spec.n_u)
        achieved = flops_unit * P * H / avg_rollout_s / 1e12
        traffic = pmc_traffic(f'cfg2 pendulum N_train={args.n_train} H={H} P={P}')
        n_mfma = pmc_mfma_instructions(f'cfg2 pendulum N_train={args.n_train} H={H} P={P}')
        executed = (n_mfma * 2048 / avg_rollout_s / 1e12) if n_mfma else None
        out = {
            'metric': 'cem_particle_step_evals_per_s', 'value': particle_steps / elapsed, 'unit': 'particle-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'cfg2 inverted pendulum n_s=2 n_u=1, exact GP N_train={args.n_train}, CEM H={H}, '
                                   f'{P} particles/GPU, {iters} CEM iterations/solve, {elites} elites; 1 step = 1 MPC solve',
                       'particles_per_gpu': P, 'horizon': H, 'n_train': args.n_train, 'cem_iterations': iters,
                       'elites': elites, 'parallelism': f'particle-sharded x{world}, 1 all-reduce/iteration'},
            'mpc_solves_per_s': args.steps / elapsed,
            'particle_rollouts_per_s': total_particles * iters * args.steps / elapsed,
            'device_status': status_word, 'solution_found': bool(ok[0].item()),
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': F64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / F64_MATRIX_PEAK_TFLOPS,
                         'traffic': traffic,
                         'traffic_unit': 'B/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_summary.json)',
                         'kernel': 'cem_rollout_kernel<2,1>',
                         'avg_launch_us': avg_rollout_s * 1e6, 'launches_timed': len(rollout_ms),
                         'hbm_gb_per_s': (traffic / avg_rollout_s / 1e9) if traffic else None,
                         'hbm_frac_of_8TBps': (traffic / avg_rollout_s / 8e12) if traffic else None,
                         'algorithmic_flops_per_launch': flops_unit * P * H,
                         # the triangular form executes about half the algorithmic flops: the matrix pipe's real load
                         'executed_tflops': executed,
                         'executed_frac': (executed / F64_MATRIX_PEAK_TFLOPS) if executed else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(spec, H, P, max(1, P // 10))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(group)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
