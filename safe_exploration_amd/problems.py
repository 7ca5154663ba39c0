"""Seeded synthetic problem instances with the constants of the reference's environments.

Used by bench.py, the smoke test and the parity tests so that the HIP path and the oracle see identical inputs.
Environment constants cite the reference (``safe_exploration/environments/environments.py``); the training sets
are synthetic (there is no simulator on the path), the GP hyper-parameters are fixed, not fitted.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import scipy.signal

from . import _lib
from .utils import dlqr


@dataclass
class ProblemSpec:
    name: str
    n_s: int
    n_u: int
    X: np.ndarray            # [N x D]
    Y: np.ndarray            # [N x n_s]  error to the linear prior
    lengthscale: np.ndarray  # [n_s x D]
    outputscale: np.ndarray  # [n_s]
    noise: np.ndarray        # [n_s]
    a: np.ndarray
    b: np.ndarray
    k_fb: np.ndarray
    l_mu: np.ndarray
    l_sigma: np.ndarray
    beta: float
    h_mat: np.ndarray
    h_vec: np.ndarray
    u_min: np.ndarray
    u_max: np.ndarray
    obj_mode: int = _lib.SX_OBJ_NEG_VARIANCE
    obj_w_abs: Optional[np.ndarray] = None
    obj_target: Optional[np.ndarray] = None
    obj_w_lin: Optional[np.ndarray] = None
    con_mode: int = _lib.SX_CON_ALL_STATES
    x0_std: float = 0.05
    extra: dict = field(default_factory=dict)


def synthetic_training_set(n: int, n_s: int, n_u: int, seed: int = 0, scale=0.5, amp: float = 0.2,
                           noise_std: float = 0.01):
    """X ~ U(-scale, scale)^D (the range of the reference's invpend_data.npz); y = the error to the linear prior, a
    smooth function of amplitude ~`amp` (sin of a random projection + amp/4 cos(3 x)) plus N(0, noise_std^2)."""
    rng = np.random.default_rng(seed)
    d_in = n_s + n_u
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), (d_in,))
    X = rng.uniform(-1.0, 1.0, size=(n, d_in)) * scale
    Wm = rng.normal(size=(d_in, n_s)) * 0.3
    Y = np.sin((X / scale) @ Wm * 0.5) * amp + 0.25 * amp * np.cos(3.0 * X[:, :n_s] / scale[:n_s]) \
        + rng.normal(size=(n, n_s)) * noise_std
    return X, Y


def _discretize(a_ct, b_ct, dt):
    n_s, n_u = b_ct.shape
    a, b, _, _, _ = scipy.signal.cont2discrete((a_ct, b_ct, np.eye(n_s), np.zeros((n_s, n_u))), dt)
    return a, b


def pendulum(n_train: int = 200, seed: int = 0, obj_mode: int = _lib.SX_OBJ_NEG_VARIANCE, beta: float = 3.0,
             simple_constraints: bool = True, model_error: float = 0.2, data_noise_std: float = 0.01,
             outputscale: float = 0.01, noise: float = 1e-5, ard: bool = False) -> ProblemSpec:
    """Inverted pendulum: state (d_theta, theta), one torque.  Constants: environments.py:403-482 (l=.5, g=9.82, dt=.05,
    u in [-1, 1], l_mu = l_sigm = [.05, .02]), polytope :779-831, prior mass .1 and LQR weights
    diag(1, 2) / 25 (experiments/journal_experiment_configs/episodic_pendulum_cem.py:32-58)."""
    n_s, n_u = 2, 1
    l, g, dt, m_prior, friction = 0.5, 9.82, 0.05, 0.1, 0.0
    inertia = m_prior * l * l
    a_ct = np.array([[-friction / inertia, g / l], [1.0, 0.0]])   # linearised at the upright position
    b_ct = np.array([[1.0 / inertia], [0.0]])
    a, b = _discretize(a_ct, b_ct, dt)
    k_fb = -dlqr(a, b, np.diag([1.0, 2.0]), 25.0 * np.eye(1))[0]
    max_rad = np.deg2rad(20.0)
    if simple_constraints:   # |d_theta| <= 0.8, |theta| <= 20 deg (the hull of (+-0.8, +-20 deg))
        h_mat = np.array([[1., 0.], [-1., 0.], [0., 1.], [0., -1.]])
        h_vec = np.array([[0.8], [0.8], [max_rad], [max_rad]])
    else:                    # the rhombus (-1.2, 20deg), (0.8, 0), (1.2, -20deg), (-0.8, 0): outward normals
        corners = np.array([[-1.2, max_rad], [0.8, 0.0], [1.2, -max_rad], [-0.8, 0.0]])
        h_rows, h_rhs = [], []
        centre = corners.mean(0)
        for i in range(4):
            p0, p1 = corners[i], corners[(i + 1) % 4]
            nrm = np.array([p1[1] - p0[1], -(p1[0] - p0[0])])
            nrm /= np.linalg.norm(nrm)
            if nrm @ (centre - p0) > 0:
                nrm = -nrm
            h_rows.append(nrm)
            h_rhs.append(nrm @ p0)
        h_mat, h_vec = np.array(h_rows), np.array(h_rhs)[:, None]
    X, Y = synthetic_training_set(n_train, n_s, n_u, seed=seed, scale=0.5, amp=model_error, noise_std=data_noise_std)
    ls, os_, nz = np.full((n_s, 3), 0.7), np.full(n_s, outputscale), np.full(n_s, noise)
    if ard:
        # What a fitted model looks like (`_train_model`, gp_ssm_cem.py:103-129): every output its own ARD length-scales,
        # outputscale and noise.  With identical hyper-parameters the two outputs' Kstar and W coincide -- the kernels do
        # not exploit that (the MFMA count is the two-output count), but a headline workload should not invite the doubt.
        ls = np.array([[0.7, 0.9, 0.6], [0.55, 0.8, 1.1]])
        os_ = outputscale * np.array([1.0, 1.6])
        nz = noise * np.array([1.0, 2.0])
    spec = ProblemSpec('pendulum', n_s, n_u, X, Y, ls, os_, nz, a, b,
                       k_fb, np.array([.05, .02]), np.array([.05, .02]), beta, h_mat, h_vec, np.array([-1.0]),
                       np.array([1.0]), obj_mode=obj_mode)
    if obj_mode == _lib.SX_OBJ_AFFINE_ABS:   # |theta_target - theta| (environments.py:505-510), first objective -0.1
        spec.obj_w_abs, spec.obj_target, spec.obj_w_lin = np.array([0., 1.]), np.array([0., -0.1]), np.zeros(2)
    return spec


def cartpole(n_train: int = 2000, seed: int = 2, beta: float = 2.0, model_error: float = 0.2,
             data_noise_std: float = 0.01, outputscale: float = 0.01, noise: float = 1e-5, l_mu=0.05,
             l_sigma=0.05) -> ProblemSpec:
    """Cart-pole: state (x, dx, theta, dtheta), one force.  Constants: environments.py:880-941 (l=.5, m=.5, M=.5, b=.1,
    dt=.1, u in [-4, 4], l_mu = l_sigm = [.05]*4), Jacobian :1049-1066, 9-row polytope :1068-1116 (un-normalised:
    defaultconfig_episode.py:61-62 sets norm_x = 1), LQR diag(2, 6, 12, 4) / 40 and beta 2 (:65-71)."""
    n_s, n_u = 4, 1
    l, m, M, fr, g, dt = 0.5, 0.55, 0.5, 0.1, 9.82, 0.1
    a_ct = np.array([[0, 1, 0, 0], [0, 0, .5 * g * m / M, -fr * .5 / (M * l)], [0, 0, 0, 1],
                     [0, 0, g * (m + M) / (l * M), -fr * (m + M) / (m * M * l ** 2)]])
    b_ct = np.array([0, 1. / M, 0, 1 / (M * l)]).reshape(-1, 1)
    a, b = _discretize(a_ct, b_ct, dt)
    k_fb = -dlqr(a, b, np.diag([2.0, 6.0, 12.0, 4.0]), 40.0 * np.eye(1))[0]
    h_mat = np.array([[0., 0., 7.25, 1.], [0., 0., -7.25, -1.], [0., 0., -1.25, -1.], [0., 0., 1.25, 1.],
                      [0., 1., 0., 0.], [0., -1., 0., 0.], [1., 0., 0., 0.], [-1., 0., 0., 0.], [1., 2., 0., 0.]])
    h_vec = np.array([1., 1., 1., 1., 1.66, 1.66, 2.6, 4.0, 3.0])[:, None]
    X, Y = synthetic_training_set(n_train, n_s, n_u, seed=seed, scale=np.array([2.0, 1.5, 0.4, 1.0, 2.0]),
                                  amp=model_error, noise_std=data_noise_std)
    ls = np.tile(np.array([[4.0, 3.0, 0.8, 2.0, 4.0]]), (n_s, 1)) * np.array([[1.0], [1.1], [0.9], [1.2]])
    return ProblemSpec('cartpole', n_s, n_u, X, Y, ls, np.full(n_s, outputscale), np.full(n_s, noise), a, b, k_fb,
                       np.full(4, float(l_mu)), np.full(4, float(l_sigma)), beta, h_mat, h_vec, np.array([-4.0]),
                       np.array([4.0]))


def build(spec: ProblemSpec, device='cuda:0'):
    """(GpCemSSM with the spec's data and hyper-parameters on `device`, sx_env)."""
    import torch

    from .gp_reachability_pytorch import make_env
    from .ssm_cem.gp_ssm_cem import GpCemSSM

    class _Conf:
        exact_gp_training_iterations = 0
        exact_gp_kernel = 'rbf'

    _Conf.device = str(device)
    ssm = GpCemSSM(_Conf(), spec.n_s, spec.n_u)
    ssm.set_hyperparameters(spec.lengthscale, spec.outputscale, spec.noise)
    ssm.update_model(torch.tensor(spec.X, dtype=torch.float64, device=device),
                     torch.tensor(spec.Y, dtype=torch.float64, device=device), replace_old=True)
    env = make_env(spec.n_s, spec.n_u, a=spec.a, b=spec.b, k_fb=spec.k_fb, l_mu=spec.l_mu, l_sigma=spec.l_sigma,
                   beta=spec.beta, h_mat=spec.h_mat, h_vec=spec.h_vec, u_min=spec.u_min, u_max=spec.u_max,
                   obj_mode=spec.obj_mode, obj_w_abs=spec.obj_w_abs, obj_target=spec.obj_target,
                   obj_w_lin=spec.obj_w_lin, con_mode=spec.con_mode)
    return ssm, env


def oracle_problem(spec: ProblemSpec, ocem):
    """The same constants as an ``oracle.cem.Problem`` (the caller passes the oracle module: this package never
    imports it)."""
    return ocem.Problem(spec.n_s, spec.n_u, spec.a, spec.b, spec.k_fb, spec.l_mu, spec.l_sigma, spec.beta, spec.h_mat,
                        spec.h_vec, spec.u_min, spec.u_max, obj_mode=spec.obj_mode, obj_w_abs=spec.obj_w_abs,
                        obj_target=spec.obj_target, obj_w_lin=spec.obj_w_lin, con_mode=spec.con_mode)


# ---------------------------------------------------------------------------------------------------------------------
# The BASELINE.json workloads, by config number.  bench.py, the parity tests and the probes all take
# their shapes and constants from here.
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class Workload:
    cfg: int
    name: str
    spec: ProblemSpec
    horizon: int
    particles: int             # per GPU (per episode for cfg 5)
    elites: int
    iterations: int
    init_std: object           # float, or one value per step [H]
    warm_start: str            # 'zero' (the reference's cold start: mean 0) or 'safe_policy' (mean = the safe controller
                               # u = k_fb x rolled through the model's mean dynamics: FusedCemMpc.safe_policy_plan)
    x0: np.ndarray             # [E x n_s] start states, one per episode
    sharded: bool = True       # particles of ONE problem shard across GPUs (False: independent episodes per GPU)
    notes: str = ''

    @property
    def episodes(self) -> int:
        return self.x0.shape[0]


def lqr_plan(spec: ProblemSpec, x0: np.ndarray, horizon: int, mean_fn=None) -> np.ndarray:
    """Open-loop action sequence [H x n_u] of the safe controller u = k_fb x (reference safempc_cem.py:259-262) rolled
    from x0 through x' = a x + b u + mean_fn(x, u); mean_fn = None: the linear prior alone.  With the GP's posterior
    mean as mean_fn this is the host-side twin of FusedCemMpc.safe_policy_plan (tests, CPU baseline)."""
    x = np.asarray(x0, dtype=np.float64).copy()
    plan = np.empty((horizon, spec.n_u))
    for t in range(horizon):
        plan[t] = spec.k_fb @ x
        x = spec.a @ x + spec.b @ plan[t] + (mean_fn(x, plan[t]) if mean_fn is not None else 0.0)
    return plan


def start_states(n_s: int, episodes: int, seed: int = 7, std: float = 0.05) -> np.ndarray:
    """x0 ~ N(0, std^2); episode 0 is the fixed state every round-1 measurement used."""
    rng = np.random.default_rng(seed)
    x0 = rng.normal(size=(episodes, n_s)) * std
    x0[0] = np.array([0.02, -0.03] + [0.0] * (n_s - 2))[:n_s]
    return x0


def baseline_workload(cfg: int, n_gpus: int = 1, n_train: Optional[int] = None, ard: bool = False) -> Workload:
    """BASELINE.json `configs[cfg - 1]` as a synthetic, seeded workload.

    Why configs 3 and 4 do not reuse config 2's GP.  One step multiplies the ellipsoid's radius by roughly
    rho(a + b k_fb) + beta l_sigma sqrt(n_s lambda_max(I + k_fb^T k_fb)) whatever the data say (the l_sigma remainder
    box, gp_reachability_pytorch.py:145-155), and adds n_s (l_mu lambda_max(Q B))^2 -- QUADRATIC in Q (:162).  With the
    reference's pendulum constants the first is 0.89 + 0.29 = 1.18 per step: harmless over the H <= 5 the reference
    runs, 1.4^30 in Q at H = 30.  A particle that leaves the data sees the prior variance; with config 2's
    outputscale 0.01 its Q then crosses the threshold 1 / (n_s l_mu^2 lambda_B^2) of the quadratic term and overflows
    float64 within the horizon -- inf / inf = NaN, and the reference raises ValueError (:149-153).  Configs 3 and 4
    therefore describe a WELL-IDENTIFIED model (error to the prior ~1e-3, GP outputscale 1e-6): every particle of
    every iteration finishes with status 0.  The cart-pole's LQR gain gives lambda_B = 342, so beta l_sigma sqrt(n_s
    lambda_B) = 3.7 per step with the environment's l_sigma = 0.05: no GP keeps that bounded for 20 steps; config 4
    takes Lipschitz constants consistent with ITS GP (l_sigma ~ sqrt(s) / l_min, l_mu ~ amplitude / l_min^2).
    """
    if cfg == 1:   # plumbing shape: pendulum, H = 5, 64 particles (the reference's CPU-runnable case; here on the GPU)
        spec = pendulum(n_train or 75, seed=0)
        return Workload(1, 'cfg1 inverted pendulum (plumbing shape)', spec, 5, 64, 8, 8, 0.1, 'zero', start_states(2, 1))
    if ard and cfg != 2:
        raise ValueError('the per-output ARD variant exists for config 2')
    if cfg == 2:
        spec = pendulum(n_train or 200, seed=0, ard=ard)
        return Workload(2, 'cfg2 inverted pendulum' + (', per-output ARD hyper-parameters' if ard else ''), spec, 15, 4096, 409, 8,
                        0.1, 'zero', start_states(2, 1),
                        notes='variant of config 2 with distinct length-scales / outputscale / noise per output' if ard else '')
    if cfg == 3:   # 65 536 particles over 8 GPUs = 8192 per GPU, H = 30
        spec = pendulum(n_train or 200, seed=0, model_error=1e-3, data_noise_std=5e-5, outputscale=1e-6, noise=2.5e-9)
        return Workload(3, 'cfg3 inverted pendulum, long horizon', spec, 30, 8192, 819, 8, 0.1, 'zero', start_states(2, 1),
                        notes='per-GPU share of the 65 536-particle problem; well-identified GP (outputscale 1e-6)')
    if cfg == 4:
        spec = cartpole(n_train or 2000, seed=2, model_error=1e-3, data_noise_std=5e-5, outputscale=1e-6, noise=2.5e-9,
                        l_mu=2e-3, l_sigma=1.5e-3)
        # The cart-pole's linear prior has an open-loop pole at 1.77 per step (dt = .1): noise on u_t reaches the end of
        # the horizon amplified by 1.77^(H-1-t), 9e4 for t = 0.  A zero-mean, constant-std start leaves no feasible
        # particle to learn from, so the workload starts the CEM at the safe controller's plan with the terminal
        # sensitivity equalised over the steps: std_t = 0.5 * 1.77^(t - (H-1)).
        lam = float(np.abs(np.linalg.eigvals(spec.a)).max())
        std = 0.5 * lam ** (np.arange(20) - 19.0)
        return Workload(4, 'cfg4 cart-pole', spec, 20, 16384, 1638, 8, std, 'safe_policy', start_states(4, 1),
                        notes='well-identified GP (outputscale 1e-6), l_mu / l_sigma consistent with it, safe-policy '
                              'warm start with std_t = 0.5 lambda^(t-H+1)')
    if cfg == 5:   # 64 independent episodes over 8 GPUs = 8 per GPU
        spec = pendulum(n_train or 200, seed=0)
        return Workload(5, 'cfg5 batched exploration', spec, 15, 4096, 409, 8, 0.1, 'zero', start_states(2, 8 * n_gpus),
                        sharded=False, notes='8 independent episodes per GPU x 4096 particles, one fused solve')
    raise ValueError(f'BASELINE.json has configs 1..5, got {cfg}')


class StubEnv:
    """The attributes and methods the solver and the lockstep runner read from an ``Environment`` (reference
    ``environments.py:44-400``), filled from a ProblemSpec.  The "true system" is the linear prior (the simulators are the
    reference's host code and stay out of scope); an episode is `done` when the state leaves the safe polytope."""

    def __init__(self, spec: ProblemSpec, x0, never_done: bool = False, objective_target: Optional[float] = None):
        self.spec = spec
        self.n_s, self.n_u = spec.n_s, spec.n_u
        self.l_mu, self.l_sigm = spec.l_mu, spec.l_sigma
        self.u_min_norm, self.u_max_norm = spec.u_min, spec.u_max
        self._x0 = np.asarray(x0, dtype=np.float64).copy()
        self._never_done = never_done
        self._current_objective = objective_target
        self.state = self._x0.copy()
        self.steps = 0

    def reset(self, mean=None, std=None):
        self.state = self._x0.copy() if mean is None else np.asarray(mean, dtype=np.float64).copy()
        self.steps = 0
        return self.state.copy()

    def random_action(self):
        return np.zeros(self.n_u)

    def step(self, action):
        """-> (applied action, next state, observation, done, env_result) like Environment.step (environments.py:95-120)."""
        action = np.clip(np.asarray(action, dtype=np.float64).reshape(self.n_u), self.spec.u_min, self.spec.u_max)
        self.state = self.spec.a @ self.state + self.spec.b @ action
        self.steps += 1
        outside = bool((self.spec.h_mat @ self.state > self.spec.h_vec.reshape(-1)).any())
        done = outside and not self._never_done
        return action, self.state.copy(), self.state.copy(), done, (1 if outside else 0)

    def objective_cost_function(self, ps):
        if self._current_objective is None:
            return None
        import torch
        return torch.abs(torch.full_like(ps[:, 1], self._current_objective) - ps[:, 1])

    def get_safety_constraints(self, normalize=True):
        return self.spec.h_mat, self.spec.h_vec, None, None

    def collect_metrics(self):
        return {'stub_env_steps': self.steps}


def make_solver(spec: ProblemSpec, conf, env=None, device='cuda:0'):
    """A ``CemSafeMPC`` over the spec's GP and constants, the way ``utils_config.create_solver``'s ``safempc_cem`` branch
    builds it (reference utils_config.py:116-121): (solver, env)."""
    from .safempc_cem import CemSafeMPC, construct_constraints
    from .ssm_cem.gp_ssm_cem import GpCemSSM
    env = env if env is not None else StubEnv(spec, np.zeros(spec.n_s))
    ssm = GpCemSSM(conf, spec.n_s, spec.n_u)
    ssm.set_hyperparameters(spec.lengthscale, spec.outputscale, spec.noise)
    q_lqr = np.diag([1.0, 2.0]) if spec.n_s == 2 else np.diag([2.0, 6.0, 12.0, 4.0])[:spec.n_s, :spec.n_s]
    r_lqr = (25.0 if spec.n_s == 2 else 40.0) * np.eye(spec.n_u)
    solver = CemSafeMPC(ssm, construct_constraints(conf, env), env, conf, {'lin_model': (spec.a, spec.b)},
                        wx_feedback_cost=q_lqr, wu_feedback_cost=r_lqr, beta_safety=spec.beta,
                        safe_policy=lambda x: spec.k_fb @ x)
    # update_model subtracts the prior, so hand it y + prior to end up with the spec's targets
    y = spec.Y + spec.X[:, :spec.n_s] @ spec.a.T + spec.X[:, spec.n_s:] @ spec.b.T
    solver.update_model(spec.X, y, opt_hyp=False, replace_old=True)
    return solver, env
