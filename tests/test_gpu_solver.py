"""GPU: the drop-in classes end to end (CemSafeMPC.get_action, the dynamics callback, the objective hook path, GP
hyper-parameter fit), checked against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cem as ocem
from oracle import reachability as oreach
from oracle.gp import ExactGP

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


class Conf:
    mpc_time_horizon = 5
    cem_num_rollouts = 200
    cem_num_elites = 20
    cem_num_iterations = 4
    cem_init_std = 0.2
    plot_cem_optimisation = False
    plot_cem_terminal_states = False
    device = DEV
    use_state_constraint = True
    use_prior_model = True
    exact_gp_training_iterations = 0
    exact_gp_kernel = 'rbf'


class Env:
    """Environment attributes the solver reads, filled from a problems.ProblemSpec."""

    def __init__(self, spec, enable_objectives, odd_hook=False):
        self.spec, self._enable, self._odd = spec, enable_objectives, odd_hook
        self.n_s, self.n_u = spec.n_s, spec.n_u
        self.l_mu, self.l_sigm = spec.l_mu, spec.l_sigma
        self.u_min_norm, self.u_max_norm = spec.u_min, spec.u_max
        self._current_objective = -0.1

    def random_action(self):
        return np.zeros(self.n_u)

    def objective_cost_function(self, ps):
        if not self._enable:
            return None
        return torch.abs(torch.full_like(ps[:, 1], self._current_objective) - ps[:, 1]) * (1.0 if not self._odd else 1.0)

    def get_safety_constraints(self, normalize=True):
        return self.spec.h_mat, self.spec.h_vec, None, None


def build_solver(enable_objectives=True, conf=Conf):
    from safe_exploration_amd import problems
    from safe_exploration_amd.safempc_cem import CemSafeMPC, construct_constraints
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    spec = problems.pendulum(n_train=120, seed=3, obj_mode=1 if enable_objectives else 0)
    env = Env(spec, enable_objectives)
    ssm = GpCemSSM(conf(), spec.n_s, spec.n_u)
    ssm.set_hyperparameters(spec.lengthscale, spec.outputscale, spec.noise)
    solver = CemSafeMPC(ssm, construct_constraints(conf(), env), env, conf(), {'lin_model': (spec.a, spec.b)},
                        wx_feedback_cost=np.diag([1.0, 2.0]), wu_feedback_cost=25.0 * np.eye(1), beta_safety=spec.beta,
                        safe_policy=lambda x: spec.k_fb @ x)
    # update_model subtracts the prior, so hand it y + prior to end up with the spec's targets
    y = spec.Y + spec.X[:, :2] @ spec.a.T + spec.X[:, 2:] @ spec.b.T
    solver.update_model(spec.X, y, opt_hyp=False, replace_old=True)
    gp = ExactGP(spec.X, ssm.y_train.cpu().numpy(), spec.lengthscale, spec.outputscale, spec.noise)
    return solver, spec, gp


@pytest.mark.parametrize('enable_objectives', [True, False])
def test_get_action_matches_oracle(enable_objectives):
    from safe_exploration_amd import problems
    from safe_exploration_amd.safempc_cem import MpcResult
    solver, spec, gp = build_solver(enable_objectives)
    np.testing.assert_allclose(solver._lqr.get_control_matrix(), spec.k_fb, rtol=1e-12)
    c = Conf
    rng = np.random.default_rng(11)
    noise = rng.normal(size=(c.cem_num_iterations, c.cem_num_rollouts, c.mpc_time_horizon, 1))
    x0 = np.array([0.01, -0.02])
    mpc = solver._solver()
    it = iter(noise)
    mpc.sample_noise = lambda episodes=1: T(next(it)[None])
    action, result = solver.get_action(x0)
    ref_best, trace = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0, noise, c.cem_num_elites,
                                     init_std=np.full((c.mpc_time_horizon, 1), c.cem_init_std))
    assert ref_best is not None and result == MpcResult.FOUND_SOLUTION
    np.testing.assert_allclose(action, ref_best[0], rtol=0, atol=1e-9)     # north_star tolerance: 1e-4
    np.testing.assert_allclose(solver._last_mpc_actions, ref_best, rtol=0, atol=1e-9)
    assert solver.x_train.shape == (120, 3)
    m, v = solver.ssm_predict(spec.X[:5])
    mo, vo, _ = gp.predict(spec.X[:5], False)
    np.testing.assert_allclose(m, mo.T, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(v, vo.T, rtol=1e-9, atol=1e-11)
    ig = solver.information_gain()
    want = [np.log(np.diag(gp.L[d])).sum() - 0.5 * gp.n * np.log(gp.noise[d]) for d in range(2)]
    np.testing.assert_allclose(ig, want, rtol=1e-9)


def test_objective_hook_path_equals_kernel_path():
    """An objective the module does not recognise is evaluated through the env hook on the recorded centres: same
    numbers as the in-kernel form of the same function."""
    solver, spec, gp = build_solver(True)
    mpc = solver._solver()
    rng = np.random.default_rng(2)
    noise = T(rng.normal(size=(Conf.cem_num_iterations, 1, Conf.cem_num_rollouts, Conf.mpc_time_horizon, 1)))
    x0 = T([[0.01, -0.02]])
    best_a, ok_a, _, _ = mpc.solve(x0, noise=noise)
    env, _ = solver._build_env()
    mpc.set_env(env, objective_hook=solver._env_objective_cost_func)
    best_b, ok_b, _, _ = mpc.solve(x0, noise=noise)
    assert int(ok_a[0]) == int(ok_b[0]) == 1
    np.testing.assert_allclose(best_a.cpu().numpy(), best_b.cpu().numpy(), rtol=0, atol=1e-12)


def test_infeasible_problem_falls_back():
    from safe_exploration_amd.safempc_cem import MpcResult

    class Tight(Conf):
        mpc_time_horizon = 12   # ellipsoids outgrow the polytope: no feasible particle
    solver, spec, gp = build_solver(False, Tight)
    action, result = solver.get_action(np.array([0.3, 0.3]))
    assert result == MpcResult.SAFE_CONTROLLER
    np.testing.assert_allclose(action, spec.k_fb @ np.array([0.3, 0.3]))


def test_dynamics_callback_matches_oracle():
    """CemSafeMPC._dynamics_func (the reference's DynamicsFunc contract, safempc_cem.py:288-302) on flat states."""
    solver, spec, gp = build_solver(False)
    rng = np.random.default_rng(4)
    P = 33
    p = rng.normal(0, 0.05, size=(P, 2))
    m = rng.normal(size=(P, 2, 2))
    q = 0.01 * (m @ m.transpose(0, 2, 1) + 0.5 * np.eye(2))
    u = rng.uniform(-0.3, 0.3, size=(P, 1))
    flat = T(np.concatenate((p, q.reshape(P, -1)), axis=1))
    nxt, cost = solver._dynamics_func(flat, T(u))
    p1, q1, sig, _ = oreach.onestep_reachability(p, gp, u, spec.l_mu, spec.l_sigma, q, spec.k_fb, spec.beta, a=spec.a,
                                                 b=spec.b)
    np.testing.assert_allclose(nxt.cpu().numpy(), oreach.pq_flatten(p1, q1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cost.cpu().numpy(), -sig.sum(1), rtol=1e-9, atol=1e-12)
    # point start: all-zero Q block means "None"
    nxt0, _ = solver._dynamics_func(T(np.concatenate((p, np.zeros((P, 4))), axis=1)), T(u))
    p1, q1, _, _ = oreach.onestep_reachability(p, gp, u, spec.l_mu, spec.l_sigma, None, spec.k_fb, spec.beta, a=spec.a,
                                               b=spec.b)
    np.testing.assert_allclose(nxt0.cpu().numpy(), oreach.pq_flatten(p1, q1), rtol=1e-9, atol=1e-12)


def test_gp_without_data_predicts_the_prior_and_training_lowers_the_loss():
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM

    class TrainConf(Conf):
        exact_gp_training_iterations = 30
    ssm = GpCemSSM(TrainConf(), 2, 1)
    m, v, j = ssm.predict_with_jacobians(T(np.zeros((3, 2))), T(np.zeros((3, 1))))
    assert m.shape == (3, 2) and j.shape == (3, 2, 3) and float(m.abs().max()) == 0.0
    np.testing.assert_allclose(v.cpu().numpy(), np.log(2) * 2 + 1e-4)        # softplus(0) + softplus(0) + floor
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, size=(60, 3))
    Y = np.stack([np.sin(2 * X[:, 0]) + 0.01 * rng.normal(size=60), X[:, 1] * X[:, 2]], 1)
    ssm.update_model(T(X), T(Y), opt_hyp=True, replace_old=True)
    losses = ssm.collect_metrics()['losses']
    assert len(losses) == 30 and losses[-1] < losses[0]
    gp = ExactGP(X, Y, ssm.lengthscale.numpy(), ssm.outputscale.numpy(), ssm.noise.numpy())
    z = rng.uniform(-1, 1, size=(10, 3))
    mo, vo, jo = gp.predict(z)
    m, v, j = ssm.predict_with_jacobians(T(z[:, :2]), T(z[:, 2:]))
    np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(v.cpu().numpy(), vo, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(j.cpu().numpy(), jo, rtol=1e-8, atol=1e-10)
    with pytest.raises(ValueError):
        ssm.predict_raw(T(np.zeros((3, 2))))                                   # wrong input width


def test_nan_status_raises_value_error():
    """A NaN at the reference's zero/NaN checks aborts the solve with ValueError (gp_reachability_pytorch.py:76-80)."""
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc
    spec = problems.pendulum(n_train=64)
    ssm, env = problems.build(spec, DEV)
    mpc = FusedCemMpc(ssm, env, 4, 64, 8, 2, device=DEV)
    flat = torch.zeros((1, 6), dtype=torch.float64, device=DEV)
    flat[0, 0] = float('nan')
    with pytest.raises(ValueError):
        mpc.get_actions(flat)
    with pytest.raises(NotImplementedError):
        mpc.get_actions(torch.ones((1, 6), dtype=torch.float64, device=DEV))   # non-point start
    with pytest.raises(ValueError):
        mpc.get_actions(torch.zeros((1, 5), dtype=torch.float64, device=DEV))


def test_batched_episodes_match_single_solves():
    """Config-5 shape in small: E independent problems in one launch == E separate solves == the oracle."""
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc
    spec = problems.pendulum(n_train=100, seed=1, obj_mode=1)
    ssm, env = problems.build(spec, DEV)
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    E, P, H, k, iters = 5, 96, 5, 10, 3
    rng = np.random.default_rng(21)
    noise = rng.normal(size=(iters, E, P, H, 1))
    x0 = rng.normal(0, 0.03, size=(E, 2))
    mpc = FusedCemMpc(ssm, env, H, P, k, iters, device=DEV, init_std=0.2)
    best, ok, _, status = mpc.solve(T(x0), noise=T(noise))
    assert int(status.item()) == 0
    for e in range(E):
        ref, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0[e], noise[:, e], k,
                                init_std=np.full((H, 1), 0.2))
        assert (ref is not None) == bool(ok[e])
        if ref is not None:
            np.testing.assert_allclose(best[e].cpu().numpy(), ref, rtol=0, atol=1e-9)
        one, ok1, _, _ = mpc.solve(T(x0[e:e + 1]), noise=T(noise[:, e:e + 1]))
        assert torch.equal(one[0], best[e]) and int(ok1[0]) == int(ok[e])


def test_cartpole_sized_rollout_vs_oracle():
    """n_s = 4 (Jacobi eigen-solve, 9-row polytope) through the fused rollout."""
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import cem_rollout
    spec = problems.cartpole(n_train=150, seed=2)
    ssm, env = problems.build(spec, DEV)
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    P, H = 70, 6
    rng = np.random.default_rng(8)
    acts = rng.normal(0, 0.6, size=(P, H, 1))
    x0 = np.array([0.05, -0.02, 0.01, 0.03])
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    ref = ocem.rollout(problems.oracle_problem(spec, ocem), gp, x0, acts)
    traj = r['traj'][0].cpu().numpy()
    np.testing.assert_allclose(traj[:, :, :4], ref.traj_p, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(traj[:, :, 4:].reshape(P, H, 4, 4), ref.traj_q, rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(r['sigma'][0].cpu().numpy(), ref.sigma, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(r['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-8, atol=1e-12)
    np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost)
    assert int(r['status'].item()) == 0 and ref.status == 0


@pytest.mark.parametrize('which,n_train,P,H,path', [('pendulum', 700, 150, 5, 'by_output'), ('cartpole', 330, 200, 4, 'by_output'),
                                                   ('cartpole', 800, 100, 4, 'by_output'), ('pendulum', 1100, 100, 4, 'big'),
                                                   ('cartpole', 1010, 64, 3, 'big')])
def test_large_training_set_path_vs_oracle(which, n_train, P, H, path):
    """Training sets whose n_s Kstar buffers do not fit in LDS together: still one launch while ONE output's Kstar fits
    (output-by-output mode of the fused kernel), the three-launch-per-step path beyond (config 4's shape in small).
    Same numbers as the oracle either way."""
    import ctypes
    from safe_exploration_amd import _lib, problems
    from safe_exploration_amd.cem_mpc import cem_rollout
    spec = getattr(problems, which)(n_train=n_train, seed=5)
    ssm, env = problems.build(spec, DEV)
    ws = _lib.lib().sx_cem_rollout_workspace_bytes(ctypes.byref(ssm.device_model), 1, P, H)
    assert (ws > 0) == (path == 'big')   # the large-N path is the one that needs a workspace (Kstar in HBM)
    small = getattr(problems, which)(n_train=200, seed=5)
    ssm_small, _ = problems.build(small, DEV)
    assert _lib.lib().sx_cem_rollout_workspace_bytes(ctypes.byref(ssm_small.device_model), 1, P, H) == 0
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    rng = np.random.default_rng(3)
    acts = rng.normal(0, 0.4 if which == 'cartpole' else 0.15, size=(P, H, spec.n_u))
    x0 = rng.normal(0, 0.02, size=spec.n_s)
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    ref = ocem.rollout(problems.oracle_problem(spec, ocem), gp, x0, acts)
    n_s = spec.n_s
    traj = r['traj'][0].cpu().numpy()
    np.testing.assert_allclose(traj[:, :, :n_s], ref.traj_p, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(traj[:, :, n_s:].reshape(P, H, n_s, n_s), ref.traj_q, rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(r['sigma'][0].cpu().numpy(), ref.sigma, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(r['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-7, atol=1e-12)
    np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost)
    assert int(r['status'].item()) == 0 and ref.status == 0
    # sampling form (mean + std * noise) and two problems at once
    mean = rng.normal(0, 0.05, size=(2, H, spec.n_u))
    std = rng.uniform(0.05, 0.2, size=(2, H, spec.n_u))
    noise = rng.normal(size=(2, 40, H, spec.n_u))
    x2 = rng.normal(0, 0.02, size=(2, n_s))
    r2 = cem_rollout(ssm, env, T(x2), H, mean=T(mean), std=T(std), noise=T(noise))
    for e in range(2):
        a_e = r2['actions'][e].cpu().numpy()
        np.testing.assert_allclose(a_e, mean[e][None] + std[e][None] * noise[e], rtol=1e-13, atol=1e-16)
        ref_e = ocem.rollout(problems.oracle_problem(spec, ocem), gp, x2[e], a_e)
        np.testing.assert_allclose(r2['obj_cost'][e].cpu().numpy(), ref_e.obj_cost, rtol=1e-7, atol=1e-12)
        np.testing.assert_array_equal(r2['con_cost'][e].cpu().numpy(), ref_e.con_cost)


@pytest.mark.parametrize('n_train', [650, 1100])   # output-by-output single launch / Kstar in HBM
def test_gp_predict_large_training_set_vs_oracle(n_train):
    from safe_exploration_amd import problems
    spec = problems.pendulum(n_train=n_train, seed=9)
    ssm, _ = problems.build(spec, DEV)
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    rng = np.random.default_rng(1)
    for P in (1, 130, 300):
        z = rng.uniform(-0.5, 0.5, size=(P, 3))
        m, v, j = ssm.predict_with_jacobians(T(z[:, :2]), T(z[:, 2:]))
        mo, vo, jo = gp.predict(z)
        np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(v.cpu().numpy(), vo, rtol=1e-7, atol=1e-11)
        np.testing.assert_allclose(j.cpu().numpy(), jo, rtol=1e-8, atol=1e-10)
        m2, v2 = ssm.predict_without_jacobians(T(z[:, :2]), T(z[:, 2:]))
        assert torch.equal(m2, m) and torch.equal(v2, v)


def test_config4_training_set_size():
    """Config 4's training-set size (cart-pole, N_train = 2000) with ROUND 1's constants (outputscale 0.01, the
    environment's l_mu / l_sigma), which only stay finite for a few steps: H = 3 against the oracle, and the full 16 384
    particles through a size-independent property (identical action sequences give identical particles wherever they
    sit in the batch).  The workload itself, at its full horizon H = 20: tests/test_workloads.py."""
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import cem_rollout
    spec = problems.cartpole(n_train=2000, seed=2)
    ssm, env = problems.build(spec, DEV)
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    P, H = 48, 3
    rng = np.random.default_rng(12)
    acts = rng.normal(0, 0.5, size=(P, H, 1))
    x0 = rng.normal(0, 0.02, size=4)
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    ref = ocem.rollout(problems.oracle_problem(spec, ocem), gp, x0, acts)
    traj = r['traj'][0].cpu().numpy()
    np.testing.assert_allclose(traj[:, :, :4], ref.traj_p, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(traj[:, :, 4:].reshape(P, H, 4, 4), ref.traj_q, rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(r['sigma'][0].cpu().numpy(), ref.sigma, rtol=1e-6, atol=1e-11)
    np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost)
    big_p = 16384
    tiled = np.tile(acts, (big_p // P + 1, 1, 1))[:big_p]
    rb = cem_rollout(ssm, env, T(x0[None]), H, actions=T(tiled[None]))
    obj = rb['obj_cost'][0].cpu().numpy()
    np.testing.assert_allclose(obj[:P], ref.obj_cost, rtol=1e-6, atol=1e-11)
    assert (obj.reshape(-1)[:(big_p // P) * P].reshape(-1, P) == obj[:P][None]).all()     # bit-identical replicas
    assert (rb['con_cost'][0].cpu().numpy()[:P] == ref.con_cost).all()


def test_numpy_state_space_model_adapter():
    """StateSpaceModel.predict (numpy in / out) over the HIP GP == oracle; __call__ gives
    (mean, var, jac_mean, jac_var) like the reference's GPyTorchSSM (ssm_pytorch/gaussian_process.py:222-231)."""
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    from safe_exploration_amd.state_space_models import HipGpStateSpaceModel
    rng = np.random.default_rng(6)
    X = rng.uniform(-1, 1, size=(80, 5))
    Y = np.stack([np.sin(X[:, 0]), X[:, 1] * 0.3, np.cos(X[:, 2]), X[:, 3] - X[:, 4]], 1) * 0.1
    ls, s, nz = rng.uniform(0.8, 2.0, size=(4, 5)), rng.uniform(0.05, 0.2, size=4), np.full(4, 1e-3)
    ssm = GpCemSSM(Conf(), 4, 1)
    ssm.set_hyperparameters(ls, s, nz)
    model = HipGpStateSpaceModel(ssm, DEV)
    model.update_model(X[:50], Y[:50], replace_old=True)
    model.update_model(X[50:], Y[50:], replace_old=False)            # merged
    gp = ExactGP(X, Y, ls, s, nz)
    z = rng.uniform(-1, 1, size=(9, 5))
    m, v, j, jv = model(z[:, :4], z[:, 4:])   # what GPyTorchSSM._predict returns with jacobians=True
    mo, vo, jo = gp.predict(z)
    assert isinstance(m, np.ndarray) and j.shape == (9, 4, 5) and jv.shape == (9, 4, 5)
    np.testing.assert_allclose(m, mo, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(v, vo, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(j, jo, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(jv, gp.variance_jacobian(z), rtol=1e-8, atol=1e-12)
    # the factor the variance Jacobian needs survives a hyper-parameter evaluation that reuses the fit workspace
    ssm.mll_and_grad(ssm.x_train, ssm.y_train)
    jv2 = ssm.predict_variance_jacobian(T(z[:, :4]), T(z[:, 4:])).cpu().numpy()
    np.testing.assert_array_equal(jv2, jv)
    m2, v2 = model.predict(z[:1, :4], z[:1, 4:])
    assert m2.shape == (1, 4) and np.array_equal(m2, m[:1])
    with pytest.raises(NotImplementedError):
        model.predict(z[:, :4], z[:, 4:], full_cov=True)
    # linearize_predict / get_reverse / get_linearize_reverse (reference ssm_pytorch/gaussian_process.py:270-336): what
    # CasadiSSMEvaluator asks of a model with linearize_mu=True (state_space_models.py:279-304)
    assert model.has_reverse and model.has_jacobian
    lm, lv, lj = model.linearize_predict(z[:, :4], z[:, 4:])
    np.testing.assert_array_equal(lm, m)
    np.testing.assert_array_equal(lj, j)
    with pytest.raises(NotImplementedError):
        model.linearize_predict(z[:, :4], z[:, 4:], jacobians=True)             # second-order outputs: single inputs only
    lm1, lv1, lj1, ljv1, hess = model.linearize_predict(z[:1, :4], z[:1, 4:], jacobians=True)
    assert hess.shape == (4, 5, 5)
    np.testing.assert_allclose(hess, gp.mean_hessian(z[:1])[0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ssm.predict_mean_hessian(T(z[:, :4]), T(z[:, 4:])).cpu().numpy(), gp.mean_hessian(z),
                               rtol=1e-9, atol=1e-12)
    seed = rng.normal(size=(2 * 4 + 4 * 5, 1))
    gs, ga = model.get_linearize_reverse(seed)
    want = jo[0].T @ seed[:4, 0] + gp.variance_jacobian(z[:1])[0].T @ seed[4:8, 0] \
        + np.einsum('dj,djl->l', seed[8:, 0].reshape(4, 5), gp.mean_hessian(z[:1])[0])
    assert gs.shape == (4, 1) and ga.shape == (1, 1)
    np.testing.assert_allclose(np.concatenate((gs, ga))[:, 0], want, rtol=1e-8, atol=1e-12)
    model.predict(z[2:5, :4], z[2:5, 4:])
    seed2 = rng.normal(size=(8, 3))
    gs2, ga2 = model.get_reverse(seed2)
    want2 = jo[2].T @ seed2[:4, 0] + gp.variance_jacobian(z[2:3])[0].T @ seed2[4:, 0]
    np.testing.assert_allclose(np.concatenate((gs2, ga2)), want2, rtol=1e-8, atol=1e-12)
    # the reverse sweep is the transpose of the forward linearisation: a finite difference of seed . outputs agrees
    eps = 1e-6
    fd = np.empty(5)
    for c in range(5):
        dz = np.zeros((1, 5)); dz[0, c] = eps
        mp, vp = model.predict((z[2:3] + dz)[:, :4], (z[2:3] + dz)[:, 4:])
        mm, vm = model.predict((z[2:3] - dz)[:, :4], (z[2:3] - dz)[:, 4:])
        fd[c] = (seed2[:4, 0] @ (mp - mm)[0] + seed2[4:, 0] @ (vp - vm)[0]) / (2 * eps)
    np.testing.assert_allclose(want2, fd, rtol=1e-5, atol=1e-8)


def test_mll_and_gradient_vs_autograd():
    """sx_gp_fit + sx_gp_mll_grad against torch autograd (CPU, float64) on the textbook formula."""
    import math
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    rng = np.random.default_rng(17)
    n, n_s, n_u = 70, 2, 1
    X = rng.uniform(-1, 1, size=(n, 3))
    Y = np.stack([np.sin(2 * X[:, 0]) + X[:, 2], X[:, 1] ** 2], 1) + 0.05 * rng.normal(size=(n, 2))
    ls = rng.uniform(0.5, 1.5, size=(2, 3))
    s = np.array([0.8, 0.3])
    nz = np.array([0.02, 0.05])
    ssm = GpCemSSM(Conf(), n_s, n_u)
    ssm.set_hyperparameters(ls, s, nz)
    mll, grad = ssm.mll_and_grad(T(X), T(Y))
    tl = torch.tensor(ls, requires_grad=True)
    ts = torch.tensor(s, requires_grad=True)
    tn = torch.tensor(nz, requires_grad=True)
    tx, ty = torch.tensor(X), torch.tensor(Y)
    vals = []
    for d in range(2):
        xs = tx / tl[d]
        sq = ((xs[:, None, :] - xs[None, :, :]) ** 2).sum(2)
        K = ts[d] * torch.exp(-0.5 * sq) + tn[d] * torch.eye(n, dtype=torch.float64)
        L = torch.linalg.cholesky(K)
        a = torch.cholesky_solve(ty[:, d:d + 1], L)[:, 0]
        vals.append(-0.5 * (ty[:, d] * a).sum() - torch.log(torch.diagonal(L)).sum() - 0.5 * n * math.log(2 * math.pi))
    torch.stack(vals).sum().backward()
    np.testing.assert_allclose(mll.numpy(), torch.stack(vals).detach().numpy(), rtol=1e-10)
    np.testing.assert_allclose(grad[:, :3].numpy(), tl.grad.numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(grad[:, 3].numpy(), ts.grad.numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(grad[:, 4].numpy(), tn.grad.numpy(), rtol=1e-8, atol=1e-10)


def test_gp_ssm_shape_contracts_and_edge_cases():
    """Mirrors the reference's shape tests for CemSSMs (test_ssm_cem.py:23-42,116-164, test_gp_ssm_cem.py:12-40)."""
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    n_s, n_u = 2, 1
    ssm = GpCemSSM(Conf(), n_s, n_u)
    assert ssm.x_train is None and ssm.y_train is None and ssm.parametric is False
    x = torch.rand((11, 3), dtype=torch.float64, device=DEV)
    y = torch.rand((11, 2), dtype=torch.float64, device=DEV)
    ssm.update_model(x, y, replace_old=True)
    assert torch.equal(ssm.x_train, x) and torch.equal(ssm.y_train, y)
    ssm.update_model(x[:4], y[:4], replace_old=False)
    assert ssm.x_train.shape == (15, 3) and ssm.y_train.shape == (15, 2)
    for n in (1, 3, 17):
        states = torch.rand((n, n_s), dtype=torch.float64, device=DEV)
        actions = torch.rand((n, n_u), dtype=torch.float64, device=DEV)
        m, v, j = ssm.predict_with_jacobians(states, actions)
        assert m.shape == (n, n_s) and v.shape == (n, n_s) and j.shape == (n, n_s, n_s + n_u)
        m, v = ssm.predict_without_jacobians(states, actions)
        assert m.shape == (n, n_s) and v.shape == (n, n_s)
        mr, vr = ssm.predict_raw(torch.cat((states, actions), dim=1))
        assert mr.shape == (n_s, n) and vr.shape == (n_s, n)
        assert not m.requires_grad and (v > 0).all()
    # empty batch
    m, v, j = ssm.predict_with_jacobians(torch.empty((0, n_s), dtype=torch.float64, device=DEV),
                                         torch.empty((0, n_u), dtype=torch.float64, device=DEV))
    assert m.shape == (0, n_s) and j.shape == (0, n_s, 3)
    # shape errors are ValueErrors, as with the reference's assert_shape
    with pytest.raises(ValueError):
        ssm.predict_with_jacobians(torch.rand((3, 3), dtype=torch.float64, device=DEV),
                                   torch.rand((3, 1), dtype=torch.float64, device=DEV))
    with pytest.raises(ValueError):
        ssm.update_model(x, y[:, :1])
    assert isinstance(ssm.collect_metrics(), dict)
    class LinConf(Conf):
        exact_gp_kernel = 'linear'

    class OddConf(Conf):
        exact_gp_kernel = 'matern'

    assert type(GpCemSSM(LinConf(), 2, 1)).__name__ == 'FeatureGpCemSSM'    # tests/test_gpu_feature_gp.py
    with pytest.raises(ValueError):                                          # the reference's own error (gp_ssm_cem.py:54)
        GpCemSSM(OddConf(), 2, 1)


def test_config3_and_config5_shapes():
    """Full-size shapes of BASELINE configs 3 and 5 on one GPU's share, through size-independent properties.
    Config 3: 65 536 particles x H=30 over 8 GPUs = 8192 particles per GPU.  Config 5: 64 episodes x 4096 particles
    striped over 8 GPUs = 8 episodes per GPU in one launch.  (Parity with the oracle at full horizon and the status
    assertions of the full-size solves: tests/test_workloads.py.)"""
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc, cem_rollout
    w3 = problems.baseline_workload(3)
    ssm3, env3 = problems.build(w3.spec, DEV)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    # config 3 share: replicated action sequences must give bit-identical particles wherever they sit in the batch
    P, H, R = w3.particles, w3.horizon, 64
    base = 0.05 * torch.randn((R, H, 1), dtype=torch.float64, device=DEV, generator=gen)
    acts = base.repeat(P // R, 1, 1).unsqueeze(0).contiguous()
    x0 = T([[0.01, -0.01]])
    r = cem_rollout(ssm3, env3, x0, H, actions=acts)
    obj = r['obj_cost'][0].view(P // R, R)
    con = r['con_cost'][0].view(P // R, R)
    assert torch.equal(obj, obj[:1].expand_as(obj)) and torch.equal(con, con[:1].expand_as(con))
    assert torch.isfinite(obj).all() and int(r['status'].item()) == 0
    # config 5 share: 8 episodes x 4096 particles in one launch == the same episodes solved one at a time
    w5 = problems.baseline_workload(5)
    ssm, env = problems.build(w5.spec, DEV)
    E, P5, H5, k, iters = 8, w5.particles, w5.horizon, w5.elites, 2
    noise = torch.randn((iters, E, P5, H5, 1), dtype=torch.float64, device=DEV, generator=gen)
    x0s = T(w5.x0[:E])
    mpc = FusedCemMpc(ssm, env, H5, P5, k, iters, device=DEV, init_std=0.1)
    best, ok, _, status = mpc.solve(x0s, noise=noise)
    for e in (0, 5):
        b1, ok1, _, _ = mpc.solve(x0s[e:e + 1].contiguous(), noise=noise[:, e:e + 1].contiguous())
        # (8 problems at once are ranked by one workgroup each, a single one by counting over the whole chip: the same
        # elite SET in a different row order, so the refit sums in a different order -- last-bit differences, where the
        # small-shape tests, which take one kernel for both, stay bit-identical)
        torch.testing.assert_close(b1[0], best[e], rtol=0, atol=1e-12)
        assert int(ok1[0]) == int(ok[e])
    assert int(status.item()) == 0


@pytest.mark.parametrize('n_s,n_u', [(1, 1), (3, 1), (2, 2), (4, 2)])
def test_other_state_action_dimensions(n_s, n_u):
    """Every (n_s, n_u) pair the library instantiates goes through GP predict, one-step reachability and the fused
    rollout against the oracle on a synthetic problem (random linear prior, LQR feedback, box polytope)."""
    from safe_exploration_amd import _lib, problems
    from safe_exploration_amd.cem_mpc import cem_rollout
    from safe_exploration_amd.utils import dlqr
    rng = np.random.default_rng(100 * n_s + n_u)
    d_in = n_s + n_u
    a = np.eye(n_s) + 0.05 * rng.normal(size=(n_s, n_s))
    b = 0.3 * rng.normal(size=(n_s, n_u))
    k_fb = -dlqr(a, b, np.eye(n_s), 5.0 * np.eye(n_u))[0]
    X, Y = problems.synthetic_training_set(77, n_s, n_u, seed=n_s * 7 + n_u, scale=0.6)
    ls = rng.uniform(0.6, 1.4, size=(n_s, d_in))
    s, nz = rng.uniform(0.01, 0.03, size=n_s), rng.uniform(1e-5, 5e-5, size=n_s)
    h_mat = np.vstack((np.eye(n_s), -np.eye(n_s)))
    h_vec = np.full((2 * n_s, 1), 0.5 if n_s <= 2 else 1.2)
    spec = problems.ProblemSpec('synthetic', n_s, n_u, X, Y, ls, s, nz, a, b, k_fb, rng.uniform(0.01, 0.05, size=n_s),
                                rng.uniform(0.01, 0.05, size=n_s), 2.5, h_mat, h_vec, np.full(n_u, -0.4),
                                np.full(n_u, 0.4), obj_mode=_lib.SX_OBJ_AFFINE_ABS, obj_w_abs=rng.uniform(0, 1, size=n_s),
                                obj_target=rng.normal(0, 0.1, size=n_s), obj_w_lin=rng.normal(0, 0.2, size=n_s))
    ssm, env = problems.build(spec, DEV)
    gp = ExactGP(X, Y, ls, s, nz)
    z = rng.uniform(-0.5, 0.5, size=(21, d_in))
    m, v, j = ssm.predict_with_jacobians(T(z[:, :n_s]), T(z[:, n_s:]))
    mo, vo, jo = gp.predict(z)
    np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(v.cpu().numpy(), vo, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(j.cpu().numpy(), jo, rtol=1e-9, atol=1e-11)
    P, H = 53, 5
    acts = rng.normal(0, 0.25, size=(P, H, n_u))
    x0 = rng.normal(0, 0.02, size=n_s)
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    ref = ocem.rollout(problems.oracle_problem(spec, ocem), gp, x0, acts)
    traj = r['traj'][0].cpu().numpy()
    np.testing.assert_allclose(traj[:, :, :n_s], ref.traj_p, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(traj[:, :, n_s:].reshape(P, H, n_s, n_s), ref.traj_q, rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(r['sigma'][0].cpu().numpy(), ref.sigma, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(r['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-8, atol=1e-11)
    np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost)
    assert int(r['status'].item()) == ref.status == 0
    assert (ref.con_cost > 0).any()
    # the refit in the rollout's prologue (sx_cem_rollout_elites) at these dimensions: few and many elite rows (the many
    # are read in several batches and again for the second pass)
    from safe_exploration_amd.cem_mpc import fused_refit_applies
    for k, H2 in ((37, 9), (1500, 6)):
        assert fused_refit_applies(ssm, 2, 48, H2)
        rows = np.concatenate([np.zeros((2, k, 2)), rng.normal(0.05, 0.2, size=(2, k, H2 * n_u))], axis=2)
        noise = rng.normal(size=(2, 48, H2, n_u))
        x02 = rng.normal(0, 0.02, size=(2, n_s))
        r1 = cem_rollout(ssm, env, T(x02), H2, elite_rows=T(rows), noise=T(noise), want_dist=True)
        for e in range(2):
            m, sd = ocem.refit(rows[e, :, 2:].reshape(k, H2, n_u))
            np.testing.assert_allclose(r1['mean'][e].cpu().numpy(), m, rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(r1['std'][e].cpu().numpy(), sd, rtol=1e-12, atol=1e-15)
        r2 = cem_rollout(ssm, env, T(x02), H2, mean=r1['mean'], std=r1['std'], noise=T(noise))
        torch.testing.assert_close(r1['actions'], r2['actions'], rtol=0, atol=0)
        torch.testing.assert_close(r1['obj_cost'], r2['obj_cost'], rtol=0, atol=0)
        torch.testing.assert_close(r1['con_cost'], r2['con_cost'], rtol=0, atol=0)


def test_batched_episodes_equal_single_solves():
    """SURVEY 8f-2 / BASELINE config 5: E episodes in one fused solve give, episode by episode, exactly what E separate
    solves give with the same noise (episodes never exchange anything)."""
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc
    spec = problems.pendulum(n_train=60)
    ssm, env = problems.build(spec, 'cuda:0')
    E, P, H, k, iters = 5, 512, 6, 40, 4
    g = torch.Generator(device='cuda:0')
    g.manual_seed(11)
    noise = torch.randn((iters, E, P, H, 1), dtype=torch.float64, device='cuda:0', generator=g)
    x0 = torch.tensor([[0.02, -0.03], [0.0, 0.0], [-0.05, 0.1], [0.1, 0.2], [0.6, 3.0]], dtype=torch.float64, device='cuda:0')
    mpc = FusedCemMpc(ssm, env, H, P, k, iters, device='cuda:0', init_std=0.2)
    best, ok, _, status = mpc.solve(x0, noise=noise)
    assert int(status.item()) == 0
    for e in range(E):
        b1, ok1, _, st1 = mpc.solve(x0[e:e + 1], noise=noise[:, e:e + 1].contiguous())
        assert int(ok1[0]) == int(ok[e])
        torch.testing.assert_close(b1[0], best[e], rtol=0, atol=0)
    # and through the reference-shaped surface: flat point states in, per-episode found flags out
    flat = torch.cat((x0, torch.zeros((E, 4), dtype=torch.float64, device='cuda:0')), dim=1)
    acts, found, _ = mpc.get_actions_batch(flat)
    assert tuple(acts.shape) == (E, H, 1) and found.dtype == torch.bool and len(found) == E
    assert bool(found[0]) and not bool(found[4])   # the last start state is far outside the safe polytope
    with pytest.raises(ValueError):
        mpc.get_actions_batch(flat[:, :5])


@pytest.mark.parametrize('n', [40, 96, 97, 128, 200, 333])
def test_fit_products_vs_oracle(n):
    """sx_gp_fit (one-workgroup kernel up to N = 96, blocked matrix-core path beyond; N on and off the 64-block grid):
    W = L^-1, alpha, sum log diag L against numpy/LAPACK."""
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, size=(n, 3))
    Y = np.stack([np.sin(2 * X[:, 0]) + X[:, 2], X[:, 1] ** 2], 1) + 0.05 * rng.normal(size=(n, 2))
    ls, s, nz = rng.uniform(0.5, 1.5, size=(2, 3)), np.array([0.8, 1.3]), np.array([2e-2, 5e-3])
    ssm = GpCemSSM(Conf(), 2, 1)
    ssm.set_hyperparameters(ls, s, nz)
    m, linv, alpha, logdet, status = ssm._fit(T(X), T(Y))
    assert int(status.item()) == 0
    gp = ExactGP(X, Y, ls, s, nz)
    np.testing.assert_allclose(np.tril(linv.cpu().numpy()), gp.linv(), rtol=1e-9, atol=1e-10)
    assert not np.triu(linv.cpu().numpy(), 1).any()          # strictly upper part: zeros
    np.testing.assert_allclose(alpha.cpu().numpy(), np.stack(gp.alpha), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(logdet.cpu().numpy(), [np.log(np.diag(L)).sum() for L in gp.L], rtol=1e-12)


def test_junk_dimensions_wrapper_over_the_hip_gp():
    """JunkDimensionsSSM(GpCemSSM) (reference utils_config.py:46-47): 2 + 1 real dimensions padded to the 4 + 2 the HIP GP
    supports == the oracle GP on the zero-padded inputs, cut back the way the reference cuts."""
    import functools
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    from safe_exploration_amd.ssm_cem.ssm_cem import JunkDimensionsSSM
    rng = np.random.default_rng(21)
    n = 70
    X = rng.uniform(-1, 1, size=(n, 3))
    Y = np.stack([np.sin(2 * X[:, 0]) + X[:, 2], X[:, 1] ** 2], 1)
    ssm = JunkDimensionsSSM(functools.partial(GpCemSSM, Conf()), state_dimen=2, action_dimen=1, junk_states=2, junk_actions=1)
    ls, s, nz = rng.uniform(0.6, 1.4, size=(4, 6)), rng.uniform(0.5, 1.0, size=4), np.full(4, 1e-3)
    ssm._ssm.set_hyperparameters(ls, s, nz)
    ssm.update_model(T(X), T(Y), replace_old=True)
    Xp = np.concatenate((X, np.zeros((n, 3))), 1)            # raw inputs: [z, all junk]   (ssm_cem.py:186-187)
    Yp = np.concatenate((Y, np.zeros((n, 2))), 1)
    gp = ExactGP(Xp, Yp, ls, s, nz)
    z = rng.uniform(-1, 1, size=(11, 3))
    mean, var, jac = ssm.predict_with_jacobians(T(z[:, :2]), T(z[:, 2:]))
    zq = np.zeros((11, 6))
    zq[:, :2], zq[:, 4:5] = z[:, :2], z[:, 2:]               # queries: [states, junk] + [actions, junk]   (:160-161)
    mo, vo, jo = gp.predict(zq)
    np.testing.assert_allclose(mean.cpu().numpy(), mo[:, :2], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(var.cpu().numpy(), vo[:, :2], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(jac.cpu().numpy(), jo[:, :2, :3], rtol=1e-9, atol=1e-12)   # leading columns, as the reference
    m2, v2 = ssm.predict_without_jacobians(T(z[:, :2]), T(z[:, 2:]))
    assert torch.equal(m2, mean) and tuple(v2.shape) == (11, 2)


@pytest.mark.parametrize('js,ja', [(2, 1), (5, 0)])
def test_get_action_through_junk_dimensions_matches_oracle(js, ja):
    """The reference's junk-dimension experiment configuration (utils_config.py:45-47: junk_state_dimen / junk_action_dimen,
    ssm_cem.py:134-210; notebooks/results.ipynb cells 14-15 time the solver against it, up to 5 junk states) through
    CemSafeMPC.get_action: the solver rolls the wrapper out step by step (kernel_family 'stepwise'), and the selected actions
    equal the oracle's CEM solve over a model that pads exactly as the reference does -- training rows [z, junk], queries
    [states, junk, actions, junk], outputs and Jacobian cut back to their leading entries.  (2, 1): the padded model is the
    inner model; (5, 0): 7 padded states are beyond the device model's limits and the wrapper folds the padding away -- same
    numbers, against the same padded oracle."""
    import functools
    from safe_exploration_amd import problems
    from safe_exploration_amd.safempc_cem import CemSafeMPC, MpcResult, construct_constraints
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM
    from safe_exploration_amd.ssm_cem.ssm_cem import JunkDimensionsSSM
    spec = problems.pendulum(n_train=90, seed=5, obj_mode=0)
    env = Env(spec, False)
    ssm = JunkDimensionsSSM(functools.partial(GpCemSSM, Conf()), state_dimen=2, action_dimen=1, junk_states=js, junk_actions=ja)
    rng = np.random.default_rng(8)
    d_pad = 3 + js + ja
    ls = np.concatenate((spec.lengthscale, rng.uniform(0.6, 1.2, size=(2, d_pad - 3))), 1)   # [n_s x d_pad] for the real outputs
    ls = np.concatenate((ls, rng.uniform(0.6, 1.2, size=(js, d_pad))), 0)                    # ... and the junk outputs
    s_out = np.concatenate((spec.outputscale, np.full(js, 0.01)))
    nz = np.concatenate((spec.noise, np.full(js, 1e-5)))
    if ssm.folded_columns is None:
        assert (js, ja) == (2, 1)
        ssm._ssm.set_hyperparameters(ls, s_out, nz)
    else:
        assert (js, ja) == (5, 0) and ssm.folded_columns == (0, 1, 2, 7) and ssm._ssm.num_actions == 2
        ssm._ssm.set_hyperparameters(ls[:2][:, list(ssm.folded_columns)], s_out[:2], nz[:2])
    solver = CemSafeMPC(ssm, construct_constraints(Conf(), env), env, Conf(), {'lin_model': (spec.a, spec.b)},
                        wx_feedback_cost=np.diag([1.0, 2.0]), wu_feedback_cost=25.0 * np.eye(1), beta_safety=spec.beta,
                        safe_policy=lambda x: spec.k_fb @ x)
    y = spec.Y + spec.X[:, :2] @ spec.a.T + spec.X[:, 2:] @ spec.b.T
    solver.update_model(spec.X, y, opt_hyp=False, replace_old=True)

    class PaddedGP:
        """the oracle's exact GP behind the reference's padding"""
        def __init__(self):
            n = spec.X.shape[0]
            self.gp = ExactGP(np.concatenate((spec.X, np.zeros((n, js + ja))), 1),
                              np.concatenate((ssm.y_train.cpu().numpy(), np.zeros((n, js))), 1), ls, s_out, nz)

        def predict(self, z, jacobians=True):
            zq = np.zeros((z.shape[0], 3 + js + ja))
            zq[:, :2], zq[:, 2 + js:3 + js] = z[:, :2], z[:, 2:]
            m, v, j = self.gp.predict(zq, jacobians)
            return m[:, :2], v[:, :2], (j[:, :2, :3] if jacobians else None)

    c = Conf
    noise = rng.normal(size=(c.cem_num_iterations, c.cem_num_rollouts, c.mpc_time_horizon, 1))
    x0 = np.array([0.01, -0.02])
    mpc = solver._solver()
    it = iter(noise)
    mpc.sample_noise = lambda episodes=1: T(next(it)[None])
    action, result = solver.get_action(x0)
    ref_best, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), PaddedGP(), x0, noise, c.cem_num_elites,
                                 init_std=np.full((c.mpc_time_horizon, 1), c.cem_init_std))
    assert ref_best is not None and result == MpcResult.FOUND_SOLUTION
    np.testing.assert_allclose(action, ref_best[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(solver._last_mpc_actions, ref_best, rtol=0, atol=1e-9)


@pytest.mark.parametrize('form', ['rh', 'rw', 'stream'])
def test_every_form_of_the_fused_rollout_vs_oracle(form):
    """The three forms of the fused rollout kernel (8 waves with W partly resident -- the default where it is
    instantiated --, 4 waves with W in the register file, W streamed from L2), each forced through SX_ROLLOUT in a process of
    its own (the library reads the switch once) and checked against the oracle on every (n_s, n_u) it is built for and on
    training-set sizes on and off the 16-row grid (tools/rw_repro.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = {'rh': ['2,1,77', '1,1,77', '2,2,77', '2,1,200', '2,1,197', '1,1,9'],
              'rw': ['2,1,77', '1,1,77', '3,1,77', '4,1,50', '4,2,40', '2,2,120', '2,1,200'],
              'stream': ['2,1,77', '4,2,60', '3,1,77', '2,1,260']}[form]
    env = dict(os.environ, SX_ROLLOUT=form, SX_ROLLOUT_STRICT='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rw_repro.py')] + shapes, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0 and 'Memory access fault' not in r.stdout + r.stderr, r.stdout[-3000:] + r.stderr[-2000:]
    want = {'rh': 'form 2', 'rw': 'form 1', 'stream': 'form 0'}[form]
    assert r.stdout.count('matches the oracle; ' + want) == len(shapes), r.stdout[-3000:]


def test_solve_fuzz_small():
    """A short run of tools/solve_fuzz.py: random small problems (training-set size, episodes, particles, horizon, elites,
    iterations, start spread), whole solves against the oracle's CEM loop with the same noise."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'solve_fuzz.py'), '25', '9'], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_rollout_fuzz_small():
    """A short run of tools/rollout_fuzz.py: GP posterior and fused rollout against the oracle over random training-set sizes
    (every residue modulo the 16-row blocks, tiny sets, both problems)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rollout_fuzz.py'), '40', '7'], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
