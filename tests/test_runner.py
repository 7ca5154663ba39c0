"""The lockstep multi-episode runner (safe_exploration_amd/episode_runner.py) against the reference's do_rollout
semantics (safe_exploration/episode_runner.py:169-354).  CPU: the solver is mocked, as the reference's own solver tests
mock the optimiser (test_safempc_cem.py:83-148).  GPU: the real CemSafeMPC over a stub environment."""
import numpy as np
import pytest

from safe_exploration_amd import problems
from safe_exploration_amd.episode_runner import do_rollout, do_rollout_batch
from safe_exploration_amd.safempc_cem import MpcResult


class ScriptedSolver:
    """get_action_batch from a script: action = 0.1 * (episode + 1), results cycle FOUND / PREVIOUS / SAFE."""

    def __init__(self):
        self.calls = []

    def reset_batch(self):
        self.calls.append('reset')

    def get_action_batch(self, states, episode_ids=None, num_episodes=None):
        self.calls.append((states.copy(), list(episode_ids), num_episodes))
        step = sum(1 for c in self.calls if c != 'reset') - 1
        order = [MpcResult.FOUND_SOLUTION, MpcResult.PREVIOUS_SOLUTION, MpcResult.SAFE_CONTROLLER]
        return (np.array([[0.1 * (e + 1)] for e in episode_ids]), [order[(step + e) % 3] for e in episode_ids])

    def collect_metrics(self):
        return {'losses': []}


class Metrics:
    def __init__(self):
        self.scalars, self.non_scalars = {}, {}

    def log_scalar(self, name, value, counter):
        self.scalars[(name, counter)] = value

    def log_non_scalars(self, d, counter):
        for k, v in d.items():
            self.non_scalars[(k, counter)] = v


def test_lockstep_runner_matches_do_rollout_semantics():
    spec = problems.pendulum(n_train=10)
    x0s = np.array([[0.01, 0.0], [0.0, 0.30], [-0.02, 0.01]])      # episode 1 starts near the 20-degree bound: it leaves
    envs = [problems.StubEnv(spec, x0) for x0 in x0s]
    solver, metrics = ScriptedSolver(), Metrics()
    res = do_rollout_batch(envs, 6, solver, metrics, episode_ids=[10, 11, 12], cost=lambda s: float(np.abs(s).sum()))
    assert solver.calls[0] == 'reset'
    # the unstable pendulum drifts out of |theta| <= 20 deg: episode 1 first; the runner keeps solving for the others only
    lengths = [r.episode_length for r in res]
    assert res[1].safety_failure and lengths[1] < 6
    live = [c[1] for c in solver.calls if c != 'reset']
    assert live[0] == [0, 1, 2] and all(c[2] == 3 for c in solver.calls if c != 'reset')
    assert any(1 not in ids for ids in live)
    for e, r in enumerate(res):
        T = lengths[e]
        assert r.exit_codes.shape == (T, 1) and len(r.cc) == T and len(r.mpc_results) == T
        # reference slicing [1:-1]: the last transition is dropped (episode_runner.py:343-345)
        assert r.xx.shape == (T - 1, 3) and r.yy.shape == (T - 1, 2)
        np.testing.assert_array_equal(r.xx[0, :2], x0s[e])
        np.testing.assert_allclose(r.xx[:, 2], 0.1 * (e + 1))
        # observation t = prior dynamics of (state t, action t); next row's state = that observation
        np.testing.assert_allclose(r.yy[0], spec.a @ x0s[e] + spec.b @ np.array([0.1 * (e + 1)]))
        if T > 2:
            np.testing.assert_array_equal(r.xx[1, :2], r.yy[0])
        want = [1.0 if m in (MpcResult.FOUND_SOLUTION, MpcResult.PREVIOUS_SOLUTION) else 0.0 for m in r.mpc_results]
        np.testing.assert_array_equal(r.exit_codes[:, 0], want)
        assert metrics.scalars[('episode_length', 10 + e)] == T
        assert metrics.scalars[('safe_controller_fallback_count', 10 + e)] == r.mpc_results.count(MpcResult.SAFE_CONTROLLER)
        assert metrics.scalars[('env_result', 10 + e)] == r.env_result
        assert ('stub_env_steps', 10 + e) in metrics.non_scalars
    # obs_frequency and the single-episode signature; no solver = random actions with exit code 5
    xx, yy, cc, codes, failed = do_rollout(problems.StubEnv(spec, x0s[0], never_done=True), 7, solver=None, obs_frequency=2)
    assert xx.shape == (3, 3) and yy.shape == (3, 2) and (codes == 5).all() and codes.shape == (7, 1) and not failed


@pytest.mark.gpu
def test_lockstep_runner_over_the_real_solver():
    """E = 4 pendulum episodes through CemSafeMPC.get_action_batch: one fused solve per step for the episodes still
    running, per-episode ladders; the trajectories follow the stub environment under the returned actions."""
    spec = problems.pendulum(n_train=120, seed=3)

    class Conf:
        mpc_time_horizon, cem_num_rollouts, cem_num_elites, cem_num_iterations, cem_init_std = 5, 256, 24, 4, 0.2
        device, use_state_constraint, use_prior_model = 'cuda:0', True, True
        exact_gp_training_iterations, exact_gp_kernel = 0, 'rbf'
        plot_cem_optimisation = plot_cem_terminal_states = False

    x0s = problems.start_states(2, 4, seed=5, std=0.03)
    envs = [problems.StubEnv(spec, x0) for x0 in x0s]
    solver, _ = problems.make_solver(spec, Conf(), envs[0])
    metrics = Metrics()
    res = do_rollout_batch(envs, 5, solver, metrics)
    assert len(res) == 4
    for e, r in enumerate(res):
        assert r.episode_length >= 1 and len(r.mpc_results) == r.episode_length
        assert set(np.unique(r.exit_codes)) <= {0.0, 1.0}
        x = x0s[e]
        for t in range(r.xx.shape[0]):
            np.testing.assert_allclose(r.xx[t, :2], x, rtol=0, atol=1e-14)
            u = r.xx[t, 2:]
            assert (u >= spec.u_min - 1e-12).all() and (u <= spec.u_max + 1e-12).all()
            x = spec.a @ x + spec.b @ u
            np.testing.assert_allclose(r.yy[t], x, rtol=0, atol=1e-14)
        assert metrics.scalars[('mpc_found_solution_count', e)] == r.mpc_results.count(MpcResult.FOUND_SOLUTION)
    # the exploration module's batched form: one fused solve for E start states
    from safe_exploration_amd.safempc_exploration import DynamicSafeMPCExploration
    expl = DynamicSafeMPCExploration(solver, envs[0])
    solver.reset_batch()
    xs, us, results = expl.find_max_variance_batch(x0s)
    assert xs.shape == (4, 2) and us.shape == (4, 1) and len(results) == 4
    x1, u1 = expl.find_max_variance(x0s[0])
    assert x1.shape == (2, 1) and u1.shape == (1, 1)
    assert expl.get_information_gain().shape == (2,) and expl.x_train.shape == (120, 3)
