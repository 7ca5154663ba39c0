"""The BASELINE.json workloads (safe_exploration_amd/problems.py:baseline_workload).

CPU part: every config is numerically ALIVE on the oracle at its full horizon (status 0, some feasible particles) --
round 1's configs 3 and 4 overflowed float64 long before H --, the oracle's GP agrees with scikit-learn's
exact GP, and bench.py's executed-flop count equals the SQ_INSTS_MFMA the profiler reported.
GPU part (-m gpu): the HIP path against the oracle at the configs' full horizons and training-set sizes, status asserted.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle
from oracle import cem as ocem
from oracle.gp import ExactGP
from safe_exploration_amd import problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first_iteration(w, P, seed=1, n_train_gp=None):
    """(oracle GP, problem, x0, actions [P x H x n_u]) of the workload's first CEM iteration."""
    spec, H = w.spec, w.horizon
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    x0 = w.x0[0]
    if w.warm_start == 'zero':
        mean = np.zeros((H, spec.n_u))
    else:
        mean = problems.lqr_plan(spec, x0, H, lambda x, u: gp.predict(np.concatenate((x, u))[None], jacobians=False)[0][0])
    std = np.broadcast_to(np.asarray(w.init_std, dtype=np.float64).reshape(-1, 1), (H, spec.n_u))
    rng = np.random.default_rng(seed)
    return gp, problems.oracle_problem(spec, ocem), x0, mean[None] + std[None] * rng.normal(size=(P, H, spec.n_u))


@pytest.mark.parametrize('cfg,P,n_train,min_feasible', [(1, 64, None, 0), (2, 2048, None, 0), (3, 2048, None, 1),
                                                        (4, 256, 300, 25), (4, 24, None, 2), (5, 1024, None, 0)])
def test_baseline_workloads_are_alive_on_the_oracle(cfg, P, n_train, min_feasible):
    """Full horizon, the workload's own start distribution: no NaN / zero-clamp / u_b <= 0 anywhere (the reference would
    raise ValueError otherwise, gp_reachability_pytorch.py:149-153), every ellipsoid finite; for the long-horizon configs
    a non-trivial share of the particles stays inside the polytope for all H steps."""
    w = problems.baseline_workload(cfg, n_train=n_train)
    gp, prob, x0, acts = first_iteration(w, P)
    r = c_oracle.rollout(prob, gp, x0, acts)
    assert r.status == 0
    assert np.isfinite(r.traj_q).all() and np.isfinite(r.traj_p).all() and np.isfinite(r.obj_cost).all()
    assert int((r.con_cost == 0).sum()) >= min_feasible
    assert acts.shape[1] == w.horizon == {1: 5, 2: 15, 3: 30, 4: 20, 5: 15}[cfg]


def test_workload_shapes_follow_baseline_json():
    with open(os.path.join(ROOT, 'BASELINE.json')) as f:
        assert len(json.load(f)['configs']) == 5
    shapes = {c: problems.baseline_workload(c, n_gpus=8) for c in range(1, 6)}
    assert (shapes[1].horizon, shapes[1].particles, shapes[1].spec.n_s) == (5, 64, 2)
    assert (shapes[2].horizon, shapes[2].particles, shapes[2].spec.X.shape) == (15, 4096, (200, 3))
    assert (shapes[3].horizon, shapes[3].particles * 8) == (30, 65536)
    assert (shapes[4].horizon, shapes[4].particles, shapes[4].spec.X.shape, shapes[4].spec.n_s) == (20, 16384, (2000, 5), 4)
    assert (shapes[5].episodes, shapes[5].particles, shapes[5].sharded) == (64, 4096, False)
    assert shapes[4].warm_start == 'safe_policy' and np.ndim(shapes[4].init_std) == 1
    # config 2 keeps round 1's problem bit for bit (its measurements stay comparable)
    old = problems.pendulum(200, seed=0)
    assert np.array_equal(old.X, shapes[2].spec.X) and np.array_equal(old.Y, shapes[2].spec.Y)
    with pytest.raises(ValueError):
        problems.baseline_workload(6)


def test_oracle_gp_agrees_with_scikit_learn():
    """gpytorch 0.3.2 is absent, so the GP's VALUES are parity-unpinned; scikit-learn's exact GP is an
    independent implementation of the same closed form and is importable here: mean and variance must agree."""
    sk = pytest.importorskip('sklearn.gaussian_process')
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    rng = np.random.default_rng(5)
    X = rng.uniform(-0.5, 0.5, size=(60, 3))
    Y = np.stack((np.sin(2 * X[:, 0]) + 0.3 * X[:, 2], np.cos(X[:, 1]) * X[:, 0]), axis=1)
    ls = np.array([[0.7, 0.5, 0.9], [0.4, 0.8, 0.6]])
    s, noise = np.array([0.5, 0.2]), np.array([1e-3, 2e-3])
    gp = ExactGP(X, Y, ls, s, noise)
    z = rng.uniform(-0.6, 0.6, size=(40, 3))
    mean, var, _ = gp.predict(z, jacobians=False)
    for d in range(2):
        ref = sk.GaussianProcessRegressor(kernel=ConstantKernel(s[d], 'fixed') * RBF(ls[d], 'fixed'), alpha=noise[d],
                                          optimizer=None).fit(X, Y[:, d])
        m, sd = ref.predict(z, return_std=True)
        np.testing.assert_allclose(mean[:, d], m, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(var[:, d] - noise[d], sd ** 2, rtol=1e-7, atol=1e-12)   # sklearn: latent variance


def test_executed_flop_count_equals_the_profiled_mfma_count():
    """bench.py prices the roofline with the MFMA instructions the kernel executes; the analytic count must be the
    SQ_INSTS_MFMA rocprofv3 measured on that launch (round 1: profiles/r01_pmc_summary.json, cfg 2)."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')) as f:
        measured = json.load(f)['per_launch_averages']['cem_rollout_kernel<2,1>']['SQ_INSTS_MFMA']
    assert bench.mfma_per_launch_fused(2, 200, 3, 4096 // 16, 15) == int(measured) == 2856960
    # the large-N kernel skips the MFMAs of a row-block beyond its diagonal: fewer than the dense tile count, more than
    # the exact triangle
    n = bench.mfma_per_launch_trmm(4, 2000, 5, 16384)
    nrb = bench.n_pad_of(2000, 5) // 16
    dense = 4 * (16384 // 16) * nrb * (2 * nrb) * 2
    tri = 4 * (16384 // 16) * nrb * (nrb + 1) * 2
    assert tri <= n < dense
    assert bench.algorithmic_flops_per_particle_step(2, 200, 3) == 168000


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
DEV = 'cuda:0'


def T(x):
    import torch
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def check_rollout(spec, r, ref, P, H, rtol_q):
    n_s = spec.n_s
    traj = r['traj'][0].cpu().numpy()
    np.testing.assert_allclose(traj[:, :, :n_s], ref.traj_p, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(traj[:, :, n_s:].reshape(P, H, n_s, n_s), ref.traj_q, rtol=rtol_q, atol=1e-12)
    np.testing.assert_allclose(r['sigma'][0].cpu().numpy(), ref.sigma, rtol=1e-6, atol=1e-14)
    np.testing.assert_allclose(r['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-6, atol=1e-14)
    np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost)
    assert int(r['status'].item()) == 0 and ref.status == 0


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,P', [(3, 96), (4, 40), (1, 64)])
def test_config_rollout_full_horizon_vs_oracle(cfg, P):
    """BASELINE configs 3 (H = 30) and 4 (cart-pole, N_train = 2000, H = 20: the three-launch path) at their FULL horizon
    and training-set size, particle count cut to what the oracle finishes in seconds; config 1's shape as it is.
    Every (p, Q), variance and cost against oracle.cem.rollout; device status 0 on both sides."""
    from safe_exploration_amd.cem_mpc import cem_rollout
    w = problems.baseline_workload(cfg)
    spec, H = w.spec, w.horizon
    gp, prob, x0, acts = first_iteration(w, P)
    ssm, env = problems.build(spec, DEV)
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    ref = c_oracle.rollout(prob, gp, x0, acts)
    check_rollout(spec, r, ref, P, H, rtol_q=1e-6)
    ref_np = ocem.rollout(prob, gp, x0, acts[:8])           # the numpy oracle (pinned on the reference's goldens) agrees
    np.testing.assert_allclose(ref_np.traj_q, ref.traj_q[:8], rtol=1e-8, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [3, 4])
def test_config_full_size_solve_is_alive(cfg):
    """The full per-GPU size of configs 3 (8192 x H=30) and 4 (16 384 x H=20, N_train = 2000) through FusedCemMpc.solve
    with the workload's own start distribution: device status 0 after every iteration, a feasible best particle, and --
    the size-independent property -- the particles of a sub-sample, re-run alone with the actions the full launch
    sampled, reproduce their costs bit for bit and agree with the oracle."""
    import torch
    from safe_exploration_amd.cem_mpc import FusedCemMpc, cem_rollout
    w = problems.baseline_workload(cfg)
    spec, H, P = w.spec, w.horizon, w.particles
    ssm, env = problems.build(spec, DEV)
    iters = 2
    mpc = FusedCemMpc(ssm, env, H, P, w.elites, iters, device=DEV, seed=3, init_std=w.init_std,
                      warm_start='safe_policy' if w.warm_start != 'zero' else 'zero', record_rollouts=False)
    x0 = T(w.x0[:1])
    best, ok, _, status = mpc.solve(x0)
    assert int(status.item()) == 0 and int(ok[0]) == 1 and bool(torch.isfinite(best).all())
    if w.warm_start != 'zero':
        gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
        plan = problems.lqr_plan(spec, w.x0[0], H, lambda x, u: gp.predict(np.concatenate((x, u))[None], jacobians=False)[0][0])
        np.testing.assert_allclose(mpc.safe_policy_plan(x0)[0].cpu().numpy(), plan, rtol=1e-7, atol=1e-12)
    # first iteration again, by hand: sample, roll out at full size, then a sub-sample alone and on the oracle
    gen = torch.Generator(device=DEV)
    gen.manual_seed(11)
    noise = torch.randn((1, P, H, spec.n_u), dtype=torch.float64, device=DEV, generator=gen)
    mean = mpc.safe_policy_plan(x0) if w.warm_start != 'zero' else torch.zeros((1, H, spec.n_u), dtype=torch.float64, device=DEV)
    std = T(np.broadcast_to(np.asarray(w.init_std, dtype=np.float64).reshape(-1, 1), (H, spec.n_u))[None])
    full = cem_rollout(ssm, env, x0, H, mean=mean.contiguous(), std=std, noise=noise)
    assert int(full['status'].item()) == 0
    con = full['con_cost'][0]
    assert bool(torch.isfinite(full['obj_cost']).all()) and int((con == 0).sum()) > 0
    pick = torch.arange(0, P, P // 32, device=DEV)[:32]
    sub_actions = full['actions'][:, pick].contiguous()
    sub = cem_rollout(ssm, env, x0, H, actions=sub_actions)
    assert torch.equal(sub['obj_cost'][0], full['obj_cost'][0, pick]) and torch.equal(sub['con_cost'][0], con[pick])
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    ref = c_oracle.rollout(problems.oracle_problem(spec, ocem), gp, w.x0[0], sub_actions[0].cpu().numpy())
    assert ref.status == 0
    np.testing.assert_allclose(sub['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-6, atol=1e-14)
    np.testing.assert_array_equal(sub['con_cost'][0].cpu().numpy(), ref.con_cost)


@pytest.mark.gpu
def test_config1_shape_get_action_vs_oracle():
    """BASELINE config 1's shape (pendulum, H = 5, 64 particles -- the reference's CPU-runnable "plumbing" case) through
    the whole boundary: CemSafeMPC.get_action against the oracle's solve with the same draws."""
    import torch
    from safe_exploration_amd.safempc_cem import CemSafeMPC, MpcResult, construct_constraints
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    w = problems.baseline_workload(1)
    spec = w.spec

    class Conf:
        mpc_time_horizon, cem_num_rollouts, cem_num_elites, cem_num_iterations = w.horizon, w.particles, w.elites, w.iterations
        cem_init_std = w.init_std
        device, use_state_constraint, use_prior_model = DEV, True, True
        exact_gp_training_iterations, exact_gp_kernel = 0, 'rbf'
        plot_cem_optimisation = plot_cem_terminal_states = False

    class Env:
        n_s, n_u = spec.n_s, spec.n_u
        l_mu, l_sigm = spec.l_mu, spec.l_sigma
        u_min_norm, u_max_norm = spec.u_min, spec.u_max

        def random_action(self):
            return np.zeros(self.n_u)

        def objective_cost_function(self, ps):
            return None

        def get_safety_constraints(self, normalize=True):
            return spec.h_mat, spec.h_vec, None, None

    ssm = GpCemSSM(Conf(), spec.n_s, spec.n_u)
    ssm.set_hyperparameters(spec.lengthscale, spec.outputscale, spec.noise)
    solver = CemSafeMPC(ssm, construct_constraints(Conf(), Env()), Env(), Conf(), {'lin_model': (spec.a, spec.b)},
                        wx_feedback_cost=np.diag([1.0, 2.0]), wu_feedback_cost=25.0 * np.eye(1), beta_safety=spec.beta,
                        safe_policy=lambda x: spec.k_fb @ x)
    solver.update_model(spec.X, spec.Y + spec.X[:, :2] @ spec.a.T + spec.X[:, 2:] @ spec.b.T, opt_hyp=False, replace_old=True)
    gp = ExactGP(spec.X, ssm.y_train.cpu().numpy(), spec.lengthscale, spec.outputscale, spec.noise)
    rng = np.random.default_rng(4)
    noise = rng.normal(size=(w.iterations, w.particles, w.horizon, 1))
    it = iter(noise)
    solver._solver().sample_noise = lambda episodes=1: T(next(it)[None])
    action, result = solver.get_action(w.x0[0])
    ref_best, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, w.x0[0], noise, w.elites,
                                 init_std=np.full((w.horizon, 1), w.init_std))
    if ref_best is None:
        assert result != MpcResult.FOUND_SOLUTION
    else:
        assert result == MpcResult.FOUND_SOLUTION
        np.testing.assert_allclose(action, ref_best[0], rtol=0, atol=1e-9)     # north_star tolerance: 1e-4
    assert solver._solver() is solver._solver()                                # sx_env is cached between calls


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with nothing exported: the parent spawns the ranks itself (it used to exit
    unless WORLD_SIZE came from outside).  Two ranks share the one card under gloo; config 1's tiny shape."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--config', '1',
                          '--steps', '3', '--warmup', '1', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                         timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['device_status'] == 0 and line['exchange_us'] > 0
    assert line['config']['baseline_config'] == 1 and line['config']['backend'] == 'gloo'
    assert 0 < line['roofline']['frac'] <= 1.0 and line['value'] > 0


@pytest.mark.gpu
def test_config3_whole_problem_on_one_gpu():
    """All 65 536 particles of config 3 on ONE GPU (round 1: refused beyond 16 384 per GPU): the ranking runs in two levels
    (4 chunks hand in their top-k rows, a second launch ranks the 4 k candidates), like the multi-GPU exchange without the
    collective.  The two-level selection equals the oracle's ranking of all candidates; the solve finishes with status 0."""
    import torch
    from safe_exploration_amd.cem_mpc import FusedCemMpc, cem_rank_refit_any, rank_chunks
    rng = np.random.default_rng(8)
    P, k, L = 40000, 500, 7
    assert rank_chunks(P) == 4
    con = rng.choice([0., 0., 3., 10., 13., 20.], size=(2, P))
    obj = rng.normal(size=(2, P))
    act = rng.normal(size=(2, P, L))
    r = cem_rank_refit_any(T(con), T(obj), T(act), k, want_rows=True)
    for e in range(2):
        want = ocem.rank(con[e], obj[e], k)
        rows = r['elite_rows'][e].cpu().numpy()
        np.testing.assert_array_equal(rows[0, 2:], act[e, want[0]])                       # the best first
        got = {tuple(v) for v in rows[:, 2:]}
        assert got == {tuple(v) for v in act[e, want]}
        mean, std = ocem.refit(act[e, want][:, :, None])
        np.testing.assert_allclose(r['mean'][e].cpu().numpy(), mean[:, 0], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(r['std'][e].cpu().numpy(), std[:, 0], rtol=1e-11, atol=1e-14)
        assert int(r['best_ok'][e]) == int(con[e, want[0]] == 0)
    w = problems.baseline_workload(3)
    ssm, env = problems.build(w.spec, DEV)
    mpc = FusedCemMpc(ssm, env, w.horizon, 65536, 2048, 2, device=DEV, seed=5, init_std=w.init_std)
    best, ok, _, status = mpc.solve(T(w.x0[:1]))
    assert int(status.item()) == 0 and int(ok[0]) == 1 and bool(torch.isfinite(best).all())
