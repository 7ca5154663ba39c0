#!/usr/bin/env python
"""Benchmark of the fused safe-MPC solve on the BASELINE.json workloads: CEM particle-step evaluations/s.

    python bench.py [--config {1,2,3,4,5}] [--gpus N] [--steps K] [--warmup W] [--backend {nccl,gloo}]

`--config` picks a BASELINE.json workload by number (default 2, the one the metric is quoted on; the constants live in
safe_exploration_amd/problems.py:baseline_workload).  `--gpus N` alone is enough: the parent process starts N fresh
children (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, nothing touches the GPU before the spawn) and
exits with the worst child's code; under `python -m torch.distributed.run ... bench.py --gpus N` the ranks are already
there and nothing is spawned.

One STEP = one complete MPC solve (`get_action`'s optimiser call): `iterations` CEM iterations, each = sample the actions,
roll all particles out for H steps (GP predict + one-step reachability + costs), rank, refit; the solver draws the standard
normals of 8 solves per generator launch, inside the timed region like everything else.  Weak scaling:
every GPU holds the workload's per-GPU particle count; configs 2-4 shard ONE problem's particles and exchange elite rows
with ONE all-gather per iteration, config 5 stripes independent episodes (no collective).  Inputs are synthetic (seeded)
and resident in HBM before the timed region.  Rank 0 prints ONE JSON line; a non-zero device status (the reference would
have raised ValueError on this workload) is an error, not a throughput.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, datasheet; DESIGN.md "Roofline"
MFMA_FLOPS = 2048               # one v_mfma_f64_16x16x4_f64 = 16 x 16 x 4 x 2 flop
# the reference's own onestep_reachability on the build container's 8 cores, cfg-2 shape (327 ms per step of
# 4096 particles; the reference cannot travel to the GPU box, so this number is quoted, not re-measured)
REFERENCE_CPU = {'value': 1.25e4, 'unit': 'particle-steps/s', 'cores': 8, 'kind': 'reference',
                 'sample': 'gp_reachability_pytorch.onestep_reachability, P=4096 N=200 f64, stand-in exact GP, torch-CPU, '
                           'measured in the build container (SURVEY.md section 6), not on this box'}
# (steps, warm-up) when no flags are given: about 2 s of GPU time each, long enough for a utilisation sampler to see the run
# (round 1's default of 100 config-2 solves was 0.13 s inside a 12 s process dominated by the CPU baseline)
DEFAULT_STEPS = {1: (2000, 50), 2: (2000, 50), 3: (500, 10), 4: (3, 1), 5: (250, 5)}


# algorithmic flops per particle-step  F = n_s [2N^2 + 2N + 2N + 2ND + 3ND + N]
def algorithmic_flops_per_particle_step(n_s, n_train, d_in):
    return n_s * (2 * n_train ** 2 + 2 * n_train + 2 * n_train + 2 * n_train * d_in + 3 * n_train * d_in + n_train)


def n_pad_of(n_train, d_in):
    return (n_train + 1 + d_in + 15) // 16 * 16          # csrc/sx_gp.hpp gp_n_pad


def mfma_per_launch_fused(n_s, n_train, d_in, tiles, horizon, totals=16):
    """v_mfma_f64_16x16x4 instructions one launch of the fused rollout kernel executes: per 16-particle tile and step, every
    output's triangular product (row-block rb = 2 (rb + 1) fragment pairs = 4 (rb + 1) MFMAs) plus `totals` column-total
    MFMAs (one per output and wave that owns row-blocks of it: waves x n_s in cem_rollout_kernel, one per wave in the
    resident forms at n_s <= the wave-group count).  (rocprofv3's SQ_INSTS_MFMA agrees to the instruction:
    profiles/*pmc*.json.)"""
    nrb = n_pad_of(n_train, d_in) // 16
    per_tile_step = n_s * 2 * nrb * (nrb + 1) + totals
    return per_tile_step * tiles * horizon


def mfma_per_launch_trmm(n_s, n_train, d_in, particles):
    """The same for ONE trmm_reduce_kernel launch (one step of the large-N path): 128 x 128 tiles, row-blocks beyond
    their diagonal skipped (csrc/sx_big.hpp)."""
    nrb = n_pad_of(n_train, d_in) // 16
    p128 = (particles + 127) // 128
    row_tiles = (nrb + 7) // 8
    per_ptile = 0
    for rt in range(row_tiles):
        rb0, rb_end = rt * 8, min(rt * 8 + 8, nrb)
        npairs = 2 * rb_end
        for rb in range(rb0, rb0 + 8):
            per_ptile += min(npairs, 2 * (rb + 1)) * 16      # 8 particle tiles x 2 MFMAs per (row-block, pair)
    return per_ptile * p128 * n_s


def mlp_mfma_per_member(widths, d_in, n_s, with_jac):
    """v_mfma_f64_16x16x4 instructions per ensemble member and 16-particle tile (csrc/sx_mlp_mfma.hpp; hidden layers padded
    to row-blocks of 16 units): layer 1 (K = D + 1 with the bias column), the 64-wide layers K-pair by K-pair, the output
    rows, and per output the reverse sweep + Jacobian rows.  SQ_INSTS_MFMA agrees (profiles/r02_pmc_mlp.json)."""
    nrb = [(w + 15) // 16 for w in widths]
    n = nrb[0] * (2 if d_in + 1 > 4 else 1)
    if len(widths) == 2:
        n += nrb[1] * 4 * nrb[0]
    n += 4 * nrb[-1]
    if with_jac:
        per_output = 4 * nrb[0] + (nrb[0] * 4 * nrb[1] if len(widths) == 2 else 0)
        n += n_s * per_output
    return n


def mlp_flops_per_particle_step(widths, d_in, n_s, n_out, members):
    """algorithmic: 2 x multiply-adds of the forward pass and of one reverse sweep per output"""
    w = [d_in] + list(widths)
    fwd = sum(a * b for a, b in zip(w[:-1], w[1:])) + w[-1] * n_out
    bwd = sum(a * b for a, b in zip(w[:-1], w[1:])) + w[-1]
    return 2 * members * (fwd + n_s * bwd)


# the kernel classes the roofline object can be about (the fused rollout of the GP / degenerate-kernel / MC-dropout models)
# (SX_PROF_ROLLOUT_FUSED, SX_PROF_TRMM_BIG, SX_PROF_ROLLOUT_FEAT, SX_PROF_ROLLOUT_MLP of include/sx_amd.h)
DOMINANT_KINDS = (0, 3, 5, 6)
DOMINANT_KERNELS = ('cem_rollout_kernel', 'trmm_reduce_kernel', 'cem_rollout_feat_kernel', 'cem_rollout_mlp_kernel')


def PROFILE_STRIDE(cfg, launches=None):
    """Every n-th launch of a kernel class carries HIP events: 16 costs < 1 % (every launch 7 %), but at least ~50
    launches must be timed -- the driver's `--steps 20` is 160 rollout launches."""
    if cfg == 4:
        return 1
    if launches is None:
        return 16
    return max(1, min(16, launches // 50))


def pmc_summary(cfg):
    """The committed rocprofv3 --pmc summary for this config (profiles/r03_pmc_cfg<N>.json, else round 2's), or None.
    bench.py cannot collect PMC counters itself; the numbers are only reported for the workload AND kernel they were
    collected on."""
    for rnd in ('r03', 'r02'):
        try:
            with open(os.path.join(ROOT, 'profiles', f'{rnd}_pmc_cfg{cfg}.json')) as f:
                return json.load(f)
        except Exception:
            continue
    return None


FORM_KERNEL = {0: 'cem_rollout_kernel', 1: 'cem_rollout_rw_kernel', 2: 'cem_rollout_rh_kernel', 3: 'cem_rollout_kernel',
               4: 'trmm_reduce_kernel'}
# column-total MFMAs per tile and step: cem_rollout_kernel 8 waves x n_s; the resident forms one per (wave, output) pair
# that owns row-blocks -- with whole wave groups per output (n_s divides the wave count) that is one per wave
def total_mfmas(form, n_s):
    waves = {1: 4, 2: 8}.get(form)
    if waves is None:
        return 8 * n_s
    return waves if waves % n_s == 0 else waves * n_s     # (upper bound for the ungrouped plans: n_s = 3)


# The card needs ~50 ms of work to reach its steady clock after the set-up's idle stretches (tools/timer_sweep.sh,
# profiles/r03_timer_sweep.txt: the same 20 timed solves read 2.5-4 % slower behind 5 warm-up solves than behind 50).  Before
# the W warm-up steps bench.py therefore runs untimed solves for this long; the JSON line says how many (`prewarm_solves`).
PREWARM_S = 0.10


def cpu_baseline(w, budget_s=12.0):
    """The oracle's C restatement (oracle/csrc, OpenMP over particles; a port of the reference's arithmetic) on the host
    cores, on a bounded sample of the same workload: chunks of particles of the first CEM iteration until `budget_s`
    seconds are spent."""
    import numpy as np
    from oracle import c_oracle
    from oracle import cem as ocem
    from oracle.gp import ExactGP
    from safe_exploration_amd import problems
    spec, H = w.spec, w.horizon
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    prob = problems.oracle_problem(spec, ocem)
    rng = np.random.default_rng(1)
    x0 = w.x0[0]
    gp_mean = lambda x, u: gp.predict(np.concatenate((x, u))[None], jacobians=False)[0][0]
    mean = problems.lqr_plan(spec, x0, H, gp_mean) if w.warm_start != 'zero' else np.zeros((H, spec.n_u))
    std = np.broadcast_to(np.asarray(w.init_std, dtype=np.float64).reshape(-1, 1)[:H], (H, spec.n_u)) \
        if np.ndim(w.init_std) else np.full((H, spec.n_u), float(w.init_std))
    threads = c_oracle.max_threads()
    c_oracle.rollout(prob, gp, x0, mean[None] + std[None] * rng.normal(size=(threads, H, spec.n_u)), want_traj=False)
    # chunk size: enough particles to keep every thread busy, small enough that one chunk stays well inside the budget
    chunk = max(threads * 4, min(w.particles, int(2e10 / (algorithmic_flops_per_particle_step(
        spec.n_s, spec.X.shape[0], spec.n_s + spec.n_u) * H)) // threads * threads or threads))
    done, t0 = 0, time.perf_counter()
    while (time.perf_counter() - t0) < budget_s:
        acts = mean[None] + std[None] * rng.normal(size=(chunk, H, spec.n_u))
        c_oracle.rollout(prob, gp, x0, acts, want_traj=False)
        done += chunk
    dt = time.perf_counter() - t0
    return {'value': done * H / dt, 'unit': 'particle-steps/s', 'cores': threads, 'kind': 'port',
            'sample': f'{done} particle rollouts (H={H}) of the same workload in chunks of {chunk}, C oracle (oracle/csrc, '
                      f'OpenMP over particles, float64), {dt:.1f} s'}


def cpu_baseline_mlp(w, ssm, budget_s=10.0):
    """--ssm mc_dropout: the numpy oracle of the ensemble (oracle.gp.DropoutEnsemble, vectorised over the particles, one
    process; numpy's BLAS threads) rolled out on chunks of particles of the first CEM iteration."""
    import numpy as np
    from oracle import cem as ocem
    from oracle.gp import DropoutEnsemble
    from safe_exploration_amd import problems
    spec, H = w.spec, w.horizon
    layers, masks = ssm.ensemble()
    model = DropoutEnsemble(layers, masks, spec.n_s, predict_std=False)
    prob = problems.oracle_problem(spec, ocem)
    rng = np.random.default_rng(1)
    std = np.full((H, spec.n_u), float(np.ravel(w.init_std)[0]))
    chunk, done, t0 = 256, 0, time.perf_counter()
    while (time.perf_counter() - t0) < budget_s:
        ocem.rollout(prob, model, w.x0[0], std[None] * rng.normal(size=(chunk, H, spec.n_u)))
        done += chunk
    dt = time.perf_counter() - t0
    return {'value': done * H / dt, 'unit': 'particle-steps/s', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': f'{done} particle rollouts (H={H}) in chunks of {chunk}, numpy oracle of the ensemble '
                      f'(oracle/gp.py DropoutEnsemble + oracle/cem.py), float64, {dt:.1f} s'}


def roofline_of(kernels, spec, n_train, d_in, E, P, H, flops_unit, cfg, mlp=None, form=0):
    """The `roofline` object of the JSON line, for the kernel that took the largest share of the timed region."""
    dominant = max(kernels, key=lambda k: kernels[k][0])
    avg_s = kernels[dominant][0] / kernels[dominant][1] * 1e-3
    if dominant == 'cem_rollout_mlp_kernel':
        # (the profiler slot of both MC-dropout rollout kernels; --ssm mc_dropout runs the matrix-core one)
        units = E * P * H
        tiles = (E * P + 15) // 16
        n_mfma = tiles * mlp['members'] * (mlp_mfma_per_member(mlp['hidden'], d_in, spec.n_s, False)
                                           + (H - 1) * mlp_mfma_per_member(mlp['hidden'], d_in, spec.n_s, True))
        dominant_name = 'cem_rollout_mlp_mfma_kernel'
    elif dominant == 'trmm_reduce_kernel':
        units = E * P                                # particle-steps one launch processes (one step of all particles)
        n_mfma = mfma_per_launch_trmm(spec.n_s, n_train, d_in, E * P)
    else:
        units = E * P * H                            # one launch = the whole rollout
        n_mfma = mfma_per_launch_fused(spec.n_s, n_train, d_in, E * ((P + 15) // 16), H, total_mfmas(form, spec.n_s))
    executed = n_mfma * MFMA_FLOPS / avg_s / 1e12
    algorithmic = flops_unit * units / avg_s / 1e12
    frac = executed / F64_MATRIX_PEAK_TFLOPS
    assert frac <= 1.0, f'roofline fraction {frac} > 1: the flop count or the timer is wrong'
    is_mlp = dominant == 'cem_rollout_mlp_kernel'
    kernel_name = dominant_name if is_mlp else (FORM_KERNEL.get(form, dominant) if dominant == 'cem_rollout_kernel' else dominant)
    pmc = pmc_summary(cfg)
    traffic = None
    if pmc and pmc.get('workload') == f'cfg{cfg} N_train={n_train} H={H} P={P} E={E}':
        by_kernel = pmc.get('hbm_traffic_bytes_per_launch', {})
        traffic = by_kernel.get(kernel_name, by_kernel.get(dominant) if kernel_name == dominant else None)
    return {'bound': 'mfma', 'achieved': executed, 'peak': F64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': frac, 'kernel': kernel_name,
            'avg_launch_us': avg_s * 1e6, 'launches_timed': kernels[dominant][1],
            'flops': 'EXECUTED: v_mfma_f64_16x16x4 instructions per launch x 2048; analytic count, equal to SQ_INSTS_MFMA ('
                     + ('csrc/sx_mlp_mfma.hpp, profiles/r02_pmc_mlp.json)' if is_mlp else 'the triangular form)'),
            'mfma_instructions_per_launch': n_mfma,
            'algorithmic_tflops': algorithmic,
            'algorithmic_flops_per_launch': flops_unit * units,
            'algorithmic_note': ('forward pass + one reverse sweep per output, 2 flops per multiply-add; the executed count '
                                 'also holds the zero padding of the 3-4 row output / Jacobian products to 16 rows' if is_mlp
                                 else 'SURVEY 8d counts 2 N^2 per output for K* Kinv; the kernel evaluates ||L^-1 k*||^2 '
                                      '(N^2), so algorithmic_tflops may exceed the peak'),
            'traffic': traffic,
            'traffic_unit': 'B/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950; '
                            'profiles/r03_pmc_cfg*.json)',
            'hbm_gb_per_s': (traffic / avg_s / 1e9) if traffic else None}


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """Parent of a plain `python bench.py --gpus N`: N fresh child processes, one per rank.  This process never touches
    the GPU (no torch import before here), so the children are ordinary process starts, not an exec after HIP init."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f'bench.py: rank(s) failed: {bad}', file=sys.stderr)
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=2, choices=[1, 2, 3, 4, 5], help='BASELINE.json config number')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one card)')
    ap.add_argument('--particles', type=int, default=0, help='per GPU; 0 = the workload\'s own')
    ap.add_argument('--horizon', type=int, default=0)
    ap.add_argument('--n-train', type=int, default=0)
    ap.add_argument('--ard', action='store_true',
                    help='config 2 with distinct per-output ARD length-scales / outputscales / noise (what a fitted model '
                         'looks like; the plain config has the same hyper-parameters for both outputs)')
    ap.add_argument('--iters', type=int, default=0, help='CEM iterations per solve (reference default 8)')
    ap.add_argument('--elites', type=int, default=0, help='0 = 10 %% of the per-GPU particle count')
    ap.add_argument('--ssm', default='gp', choices=['gp', 'mc_dropout'],
                    help='state-space model: the exact GP of the BASELINE configs, or (not a BASELINE config) the '
                         'reference\'s default MC-dropout network, 64 x 64 hidden units, 30 members, on the same problem')
    ap.add_argument('--rccl-one-rank', action='store_true',
                    help='with --gpus 1: run the SHARDED code path (local ranking, RCCL collective, global ranking) over an nccl '
                         'group of one rank -- the fixed cost of the exchange without any inter-GPU latency')
    ap.add_argument('--no-exchange-timer', action='store_true', help='no events around the multi-GPU exchange (exchange_us null)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timer', action='store_true', help='leave the per-launch HIP events off (no roofline object)')
    ap.add_argument('--profile-stride', type=int, default=0,
                    help='bracket every n-th launch of a kernel with HIP events (0 = PROFILE_STRIDE: 16, less in short runs)')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries ONE JSON line and nothing else: whatever the libraries below print to fd 1 (gloo's connection
    # banner, for one) goes to stderr; the line itself is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from safe_exploration_amd import _lib, problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc, fold_status

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: safe_exploration_amd has no CPU path')
    n_dev = torch.cuda.device_count()
    dev = torch.device('cuda', local_rank % n_dev)    # (several ranks may share a card under --backend gloo)
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
        group = dist.group.WORLD
    force_exchange = False
    if args.rccl_one_rank:
        if world != 1 or args.config == 5:
            raise SystemExit('--rccl-one-rank goes with --gpus 1 and a sharded workload (configs 1-4)')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        group, force_exchange = dist.group.WORLD, True

    w = problems.baseline_workload(args.config, n_gpus=world, n_train=args.n_train or None, ard=args.ard)
    spec = w.spec
    P = args.particles or w.particles
    H = args.horizon or w.horizon
    iters = args.iters or w.iterations
    elites = min(args.elites or max(1, P // 10), 2048)
    steps, warmup = DEFAULT_STEPS[w.cfg]
    steps = args.steps if args.steps is not None else steps
    warmup = args.warmup if args.warmup is not None else warmup
    n_train, d_in = spec.X.shape[0], spec.n_s + spec.n_u
    ssm, env = problems.build(spec, dev)
    mlp = None
    if args.ssm == 'mc_dropout':
        if w.cfg == 5:
            raise SystemExit('--ssm mc_dropout: configs 1-4 (config 5 drives the GP solver through the episode runner)')
        from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
        mlp = {'hidden': [64, 64], 'members': 30}

        class DropConf:         # experiments/sacred_helper.py:96-106 (network and member count), a short training
            mc_dropout_training_iterations, mc_dropout_num_samples, mc_dropout_predict_std = 500, mlp['members'], False
            mc_dropout_reinitialize, mc_dropout_hidden_features, mc_dropout_type = False, mlp['hidden'], 'fixed'
            mc_dropout_fixed_probability, mc_dropout_on_input, mc_dropout_lengthscale, device = 0.02, False, 1e-4, str(dev)

        ssm = McDropoutSSM(DropConf(), spec.n_s, spec.n_u)
        ssm.update_model(torch.tensor(spec.X, dtype=torch.float64, device=dev),
                         torch.tensor(spec.Y, dtype=torch.float64, device=dev), replace_old=True)
    if w.sharded:
        # ONE problem, particles sharded over the GPUs; the elite count does not grow with the GPU count: every rank
        # contributes its local top-k rows, so the per-iteration all-gather stays at G x (k + 1) x (2 + H n_u) doubles
        E, total_particles, solver_group = 1, P * world, group
        x0 = torch.tensor(w.x0[:1], dtype=torch.float64, device=dev)
    else:
        # independent episodes striped over the GPUs ("replicas only"): no data-path collective
        per_gpu = w.episodes // world
        E, total_particles, solver_group = per_gpu, P, None
        x0 = torch.tensor(w.x0[rank * per_gpu:(rank + 1) * per_gpu], dtype=torch.float64, device=dev)
    init_std = w.init_std if np.ndim(w.init_std) == 0 else np.asarray(w.init_std)[:H]
    lib = _lib.lib()

    def barrier():
        if world > 1:
            dist.barrier(group)
        torch.cuda.synchronize(dev)

    def prewarm(one_step):
        # untimed: the card to its steady clock (PREWARM_S above) before the W warm-up steps.  The SAME number of steps on
        # every rank (a sharded step holds a collective): the second step is timed, the slowest rank's time decides
        one_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        one_step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            dt = float(t.item())
        n = max(1, min(int(PREWARM_S / max(dt, 1e-6)), 2000))
        for _ in range(n):
            one_step()
        torch.cuda.synchronize(dev)
        return n + 2

    def start_timer():
        if not args.no_kernel_timer:
            # HIP events on every 16th launch of each kernel (every launch of the large-N path, whose kernels run for
            # milliseconds): timing every 130 us launch costs the solve ~7 % (measured), every 3rd 2.4 %, every 16th < 1 %
            # (profiles/r03_timer_sweep.txt).  The dominant kernel: at least ~50 timed launches however short the run; the
            # others every 16th at most
            dominant = args.profile_stride or PROFILE_STRIDE(w.cfg, steps * iters)
            _lib.check(lib.sx_profile_stride(max(dominant, 1 if w.cfg == 4 else 16)), 'sx_profile_stride')
            for kind in DOMINANT_KINDS:
                _lib.check(lib.sx_profile_stride_kind(kind, dominant), 'sx_profile_stride_kind')
            _lib.check(lib.sx_profile_enable(max(4096, steps * iters * (3 * H + 4))), 'sx_profile_enable')

    if w.cfg == 5:
        # config 5 runs through the product's own multi-episode driver: E episodes in lockstep over stub environments
        # (episode_runner.do_rollout_batch -> CemSafeMPC.get_action_batch -> one fused solve per step, the per-episode
        # fallback ladders and the one device->host hand-off per step included); 1 step = 1 lockstep step
        from safe_exploration_amd.episode_runner import do_rollout_batch

        class Conf:
            mpc_time_horizon, cem_num_rollouts, cem_num_elites, cem_num_iterations, cem_init_std = H, P, elites, iters, init_std
            device, use_state_constraint, use_prior_model = str(dev), True, True
            exact_gp_training_iterations, exact_gp_kernel, cem_seed = 0, 'rbf', 1
            plot_cem_optimisation = plot_cem_terminal_states = False

        envs = [problems.StubEnv(spec, x, never_done=True) for x in x0.cpu().numpy()]
        solver, _ = problems.make_solver(spec, Conf(), envs[0], dev)
        mpc = solver._solver()
        prewarm_solves = prewarm(lambda: do_rollout_batch(envs, 1, solver))
        do_rollout_batch(envs, warmup, solver)
        start_timer()
        barrier()
        t0 = time.perf_counter()
        episodes = do_rollout_batch(envs, steps, solver)
        barrier()
        elapsed = time.perf_counter() - t0
        status = torch.tensor([mpc.last_status], dtype=torch.int32)
        from safe_exploration_amd.safempc_cem import MpcResult
        ok = torch.tensor([int(any(MpcResult.FOUND_SOLUTION in r.mpc_results for r in episodes))])
        assert all(r.episode_length == steps for r in episodes)
    else:
        mpc = FusedCemMpc(ssm, env, H, total_particles, elites, iters, device=dev, seed=1, init_std=init_std,
                          warm_start='safe_policy' if w.warm_start != 'zero' else 'zero', process_group=solver_group,
                          force_exchange=force_exchange)
        prewarm_solves = prewarm(lambda: mpc.solve(x0))
        for _ in range(warmup):
            mpc.solve(x0)
        start_timer()
        if ((world > 1 and w.sharded) or force_exchange) and not args.no_exchange_timer:
            mpc.exchange_events = []
        barrier()
        t0 = time.perf_counter()
        statuses = []
        for _ in range(steps):
            best, ok, _, status = mpc.solve(x0)
            statuses.append(status)          # (a view into the solver's status pool: no launch, no synchronisation)
        barrier()
        elapsed = time.perf_counter() - t0
        status = torch.cat([t.reshape(-1) for t in statuses])   # EVERY timed solve's status word is checked below
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    status_word = fold_status(status.cpu())
    kernels = {}
    if not args.no_kernel_timer:
        kernels = _lib.profile_collect()             # {kernel: (total ms, launches)}, HIP events on the launch stream
        _lib.check(lib.sx_profile_disable(), 'sx_profile_disable')
    # The synchronous call a control loop makes (episode_runner.py:216-221 times solver.get_action): state in, one
    # device->host hand-off with status, flag and actions out -- timed after the kernel timer is off, on a bounded number of
    # solves; `value` stays the enqueue-only throughput of the region above.
    sync_ms = None
    if w.cfg != 5 and status_word == 0:
        n_s = spec.n_s
        flat = torch.cat((x0[:1], torch.zeros((1, n_s * n_s), dtype=torch.float64, device=dev)), dim=1)
        n_sync = max(1, min(steps, 200))
        mpc.get_actions(flat)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_sync):
            mpc.get_actions(flat)
        barrier()
        sync_ms = (time.perf_counter() - t1) / n_sync * 1e3
        if world > 1:
            t = torch.tensor([sync_ms], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            sync_ms = float(t.item())
    exchange_us = exchange_parts = None
    if mpc.exchange_events:
        # per sharded iteration (every 16th is bracketed): local ranking | collective | global ranking; exchange_us = the two
        # multi-GPU parts together, as in round 2
        n = len(mpc.exchange_events)
        part = lambda i: sum(ev[i].elapsed_time(ev[i + 1]) for ev in mpc.exchange_events) / n * 1e3
        exchange_parts = {'local_rank_us': part(0), 'collective_us': part(1), 'global_rank_us': part(2),
                          'iterations_timed': n,
                          'collective': ('ncclAllGather on the compute stream (own RCCL communicator, distributed.RcclComm)'
                                         if getattr(mpc, '_comm', None) is not None else 'torch.distributed ' + str(args.backend))}
        exchange_us = exchange_parts['collective_us'] + exchange_parts['global_rank_us']
    if world > 1 and not w.sharded:
        sw = torch.tensor([status_word], dtype=torch.int32, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(sw, op=dist.ReduceOp.MAX, group=group)    # (bit-OR would do; any non-zero word fails the run)
        status_word = int(sw.item())

    rc = 0
    if rank == 0:
        if status_word != 0:
            print(f'bench.py: device status {status_word} (SX_STATUS_* bits) on {w.name}: the reference would have raised '
                  f'on this workload; no throughput reported', file=sys.stderr)
            rc = 3
        else:
            episodes_total = E * (world if not w.sharded else 1)
            particle_steps = (P * world if w.sharded else P * episodes_total) * H * iters * steps
            flops_unit = algorithmic_flops_per_particle_step(spec.n_s, n_train, d_in) if mlp is None else \
                mlp_flops_per_particle_step(mlp['hidden'], d_in, spec.n_s, spec.n_s, mlp['members'])
            # every stride-th launch of a kernel class is timed (start_timer)
            dominant = args.profile_stride or PROFILE_STRIDE(w.cfg, steps * iters)
            stride_of = lambda k: dominant if k in DOMINANT_KERNELS else max(dominant, 1 if w.cfg == 4 else 16)
            form = int(lib.sx_cem_rollout_form(ctypes.byref(ssm.device_model), H)) if mlp is None else 0
            # the events are attached to the launch (sx_launch.hpp) and read the dispatch's own begin -> end, so the shares
            # need no correction for marker packets (round 2's hipEventRecord pairs read 2.8 us high and summed to 1.017);
            # the sum is still capped at 1
            per_kernel = {}
            for k, (ms, n) in kernels.items():
                per_kernel[FORM_KERNEL.get(form, k) if k == 'cem_rollout_kernel' else k] = {
                    'avg_launch_us': ms / n * 1e3, 'launches_timed': n, 'every_nth_launch': stride_of(k),
                    'share_of_step': ms * stride_of(k) / (elapsed * 1e3)}
            total_share = sum(v['share_of_step'] for v in per_kernel.values())
            if total_share > 1.0:
                for v in per_kernel.values():
                    v['share_of_step'] /= total_share
            roofline = None
            if kernels:
                roofline = roofline_of(kernels, spec, n_train, d_in, E, P, H, flops_unit, w.cfg, mlp, form)
            out = {
                'metric': 'cem_particle_step_evals_per_s', 'value': particle_steps / elapsed, 'unit': 'particle-steps/s',
                'n_gpus': world, 'steps': steps, 'warmup': warmup, 'prewarm_solves': prewarm_solves,
                'ms_per_step': elapsed / steps * 1e3,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': f'{w.name}: n_s={spec.n_s} n_u={spec.n_u}, '
                                       + (f'exact GP N_train={n_train}' if mlp is None else
                                          f'MC-dropout ensemble {mlp["hidden"]} x {mlp["members"]} members (NOT a BASELINE '
                                          f'config: the reference\'s other state-space model on the same problem)')
                                       + f', CEM H={H}, '
                                       f'{P} particles/GPU' + (f' x {E} episodes/GPU' if not w.sharded else '')
                                       + f', {iters} CEM iterations/solve, {elites} elites; 1 step = 1 MPC solve'
                                       + (f' ({w.notes})' if w.notes else ''),
                           'baseline_config': w.cfg, 'particles_per_gpu': P, 'episodes_per_gpu': E, 'horizon': H,
                           'n_train': n_train, 'cem_iterations': iters, 'elites': elites, 'warm_start': w.warm_start,
                           'ranking': ('cem_rank_count_kernel (counting, whole chip)' if int(lib.sx_cem_rank_counts(E, P)) == 1
                                       else 'cem_rank_kernel (one workgroup per problem)') + '; timed as cem_rank_kernel',
                           'parallelism': (f'particle-sharded x{world}, 1 all-gather/iteration' if w.sharded
                                           else f'episodes striped x{world}, no collective; lockstep runner '
                                                f'(episode_runner.do_rollout_batch)'),
                           'backend': args.backend if world > 1 else None},
                'mpc_solves_per_s': steps * episodes_total / elapsed,
                'sync_solve_ms': sync_ms,
                'get_action_solves_per_s': (1e3 / sync_ms) if sync_ms else None,
                'timed_solves_status_checked': int(status.numel()),
                'particle_rollouts_per_s': particle_steps / H / elapsed,
                'device_status': status_word, 'solution_found': bool(ok[0].item()),
                'exchange_us': exchange_us,
                'exchange_breakdown': exchange_parts,
                'roofline': roofline,
                'kernels': per_kernel,
            }
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline(w) if mlp is None else cpu_baseline_mlp(w, ssm)
                if w.cfg == 2 and mlp is None:
                    out['cpu_baseline']['reference_measured_elsewhere'] = REFERENCE_CPU
            os.write(json_fd, (json.dumps(out) + '\n').encode())
    if world > 1 or force_exchange:
        dist.barrier(group)
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == '__main__':
    main()
