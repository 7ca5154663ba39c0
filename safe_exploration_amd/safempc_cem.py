"""Safe MPC with the constrained cross-entropy optimiser, solved on the GPU by libsxamd.

Drop-in for the reference's ``safe_exploration/safempc_cem.py``: same class names, constructor signature, members and
fallback ladder (``CemSafeMPC.get_action`` :231-263), so ``utils_config.create_solver``'s ``safempc_cem`` branch and the
task runners work unchanged.  What differs is underneath: instead of handing Python callbacks to the third-party
``ConstrainedCemMpc`` (H sequential dynamics calls + one constraint call per trajectory per iteration), the solver hands
the GP operands and the environment constants to ``FusedCemMpc`` (two kernel launches per CEM iteration).
"""
from enum import Enum
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from numpy import ndarray
from torch import Tensor

from . import _lib, gp_reachability_pytorch
from .cem_mpc import FusedCemMpc, Rollouts
from .gp_reachability_pytorch import make_env, onestep_reachability
from .safempc import SafeMPC
from .ssm_cem.gp_ssm_cem import GpCemSSM
from .ssm_cem.ssm_cem import CemSSM
from .utils import assert_shape, dlqr, get_device


class MpcResult(Enum):
    """How get_action arrived at the action it returned (reference safempc_cem.py:23-27)."""
    FOUND_SOLUTION = 1
    PREVIOUS_SOLUTION = 2
    SAFE_CONTROLLER = 3


class PQFlattener:
    """(p [N x n_s], Q [N x n_s x n_s]) <-> flat [N x (n_s + n_s^2)], Q row-major; an all-zero Q block over the whole
    batch stands for "every state is a point" (reference safempc_cem.py:30-76)."""

    def __init__(self, state_dimen: int):
        self._n = state_dimen

    def get_flat_state_dimen(self) -> int:
        return self._n + self._n * self._n

    def flatten(self, p: Tensor, q: Optional[Tensor]) -> Tensor:
        n_batch = p.size(0)
        if q is None:
            q = torch.zeros((n_batch, self._n, self._n), dtype=p.dtype, device=p.device)
        assert_shape(p, (n_batch, self._n))
        assert_shape(q, (n_batch, self._n, self._n))
        return torch.cat((p.reshape(n_batch, -1), q.reshape(n_batch, -1)), dim=1)

    def unflatten(self, flat: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
        n_batch = flat.size(0)
        assert_shape(flat, (n_batch, self.get_flat_state_dimen()))
        p = flat[:, :self._n]
        q = flat[:, self._n:].reshape(n_batch, self._n, self._n)
        if not bool((q != 0).any()):
            q = None
        return p, q


class ActionConstraint:
    """Box constraint on the actions; costs 3 per violating time step (reference test_safempc_cem.py:59-71)."""
    COST = 3.0

    def __init__(self, u_min, u_max):
        self.u_min = np.asarray(u_min, dtype=np.float64).reshape(-1)
        self.u_max = np.asarray(u_max, dtype=np.float64).reshape(-1)

    def __call__(self, trajectory: Tensor, actions: Tensor) -> float:
        lo = torch.as_tensor(self.u_min, dtype=actions.dtype, device=actions.device)
        hi = torch.as_tensor(self.u_max, dtype=actions.dtype, device=actions.device)
        bad = ((actions < lo) | (actions > hi)).any(dim=-1)
        return float(bad.sum().item()) * self.COST


def box2torchpoly(box) -> Tuple[np.ndarray, np.ndarray]:
    """[[lo, hi], ...] -> (u_min, u_max); stands in for constrained_cem_mpc.box2torchpoly at the reference call site
    safempc_cem.py:139."""
    box = np.asarray(box, dtype=np.float64)
    return box[:, 0], box[:, 1]


class EllipsoidStateConstraint:
    """Every state ellipsoid must lie in the polytope ``a x <= b``; cost 10 per ellipsoid that does not
    (reference safempc_cem.py:116-132).  Like the reference it looks at the LAST row of the trajectory it is given."""
    COST = 10.0
    mode = _lib.SX_CON_ALL_STATES

    def __init__(self, state_dimen: int, safe_polytope_a: np.ndarray, safe_polytope_b: np.ndarray, device: str):
        self._pq = PQFlattener(state_dimen)
        self.polytope_a = np.asarray(safe_polytope_a, dtype=np.float64)
        self.polytope_b = np.asarray(safe_polytope_b, dtype=np.float64).reshape(-1, 1)
        self._device = device
        self._polytope_a = torch.tensor(self.polytope_a, device=device)
        self._polytope_b = torch.tensor(self.polytope_b, device=device)

    def __call__(self, trajectory: Tensor, actions: Tensor) -> float:
        p, q = self._pq.unflatten(trajectory.unsqueeze(0)[:, -1])
        if q is None:
            q = torch.zeros((p.size(0), p.size(1), p.size(1)), dtype=p.dtype, device=p.device)
        inside = gp_reachability_pytorch.is_ellipsoid_inside_polytope(p.contiguous(), q.contiguous(), self._polytope_a,
                                                                      self._polytope_b)
        return float((inside.size(0) - inside.sum()).item()) * self.COST


class EllipsoidTerminalConstraint:
    """Only the terminal ellipsoid must lie in the polytope (reference safempc_cem.py:102-113)."""
    mode = _lib.SX_CON_TERMINAL

    def __init__(self, state_dimen: int, safe_polytope_a: np.ndarray, safe_polytope_b: np.ndarray, device: str):
        self._constraint = EllipsoidStateConstraint(state_dimen, safe_polytope_a, safe_polytope_b, device)
        self.polytope_a = self._constraint.polytope_a
        self.polytope_b = self._constraint.polytope_b
        self._polytope_a = self._constraint._polytope_a
        self._polytope_b = self._constraint._polytope_b

    def __call__(self, trajectory: Tensor, actions: Tensor) -> float:
        return self._constraint(trajectory[-1:, :], actions[-1:, :])


def construct_constraints(conf, env):
    """[action box, state/terminal polytope] from the environment (reference safempc_cem.py:135-146)."""
    h_mat_safe, h_safe, _, _ = env.get_safety_constraints(normalize=True)
    action_constraint = ActionConstraint(*box2torchpoly(np.array([np.array(xs) for xs in zip(env.u_min_norm,
                                                                                            env.u_max_norm)])))
    cls = EllipsoidStateConstraint if conf.use_state_constraint else EllipsoidTerminalConstraint
    return [action_constraint, cls(env.n_s, h_mat_safe, h_safe, get_device(conf))]


class LqrFeedbackController:
    """k_fb = -K_lqr for the linear prior, computed once on the host (reference safempc_simple.py:1105-1129)."""

    def __init__(self, wx_feedback_cost, wu_feedback_cost, n_s: int, n_u: int, linearized_model_a, linearized_model_b,
                 conf=None):
        self._device = get_device(conf)
        self._args = (linearized_model_a, linearized_model_b, wx_feedback_cost, wu_feedback_cost)
        self._k_fb = None

    def get_control_matrix(self) -> np.ndarray:
        if self._k_fb is None:
            self._k_fb = -dlqr(*self._args)[0]
        return self._k_fb

    def get_control_matrix_pytorch(self) -> Tensor:
        return torch.tensor(self.get_control_matrix(), device=self._device)


def objective_spec(env) -> Optional[Tuple[int, np.ndarray, np.ndarray, np.ndarray]]:
    """Maps the environment's objective hook (reference environments.py:149-156) onto the forms the rollout kernel
    evaluates: (mode, w_abs, target, w_lin).  Returns None for a hook this module does not recognise (the solver then
    evaluates the hook itself on the recorded trajectory centres)."""
    n_s = env.n_s
    zeros = np.zeros(n_s)
    probe = torch.zeros((1, n_s), dtype=torch.float64)
    if env.objective_cost_function(probe) is None:          # default: maximise the predicted variance
        return _lib.SX_OBJ_NEG_VARIANCE, zeros, zeros, zeros
    # Any hook of the separable form  sum_i w_i |t_i - p_i| + v . p  -- the pendulum's |theta_target - theta|
    # (environments.py:505-510), the lunar lander's -height (lunarlander.py:111-113) -- is identified by probing the public
    # hook alone: far from the kinks the cost along axis i is a line on either side, with slopes v_i -+ w_i that meet at t_i.
    hook = lambda x: np.asarray(env.objective_cost_function(torch.as_tensor(x, dtype=torch.float64)), dtype=np.float64).reshape(-1)
    w_abs, target, w_lin = np.zeros(n_s), np.zeros(n_s), np.zeros(n_s)
    for i in range(n_s):
        far = 1024.0
        for _ in range(2):       # second pass: probes just outside the kink found by the first (rounding ~ far * 2^-53)
            x = np.zeros((4, n_s))
            x[:, i] = [-2 * far, -far, far, 2 * far]
            c = hook(x)
            if not np.all(np.isfinite(c)):
                return None          # a hook that is not finite far out is not of this form: evaluate it as given
            s_lo, s_hi = (c[1] - c[0]) / far, (c[3] - c[2]) / far
            w_abs[i], w_lin[i] = 0.5 * (s_hi - s_lo), 0.5 * (s_hi + s_lo)
            if abs(w_abs[i]) <= 1e-12:
                w_abs[i] = 0.0
                break
            # the two lines c_lo + s_lo x and c_hi + s_hi x cross at the kink
            target[i] = ((c[1] + s_lo * far) - (c[2] - s_hi * far)) / (s_hi - s_lo)
            far = float(2.0 ** np.ceil(np.log2(2.0 * abs(target[i]) + 2.0)))
        if abs(w_lin[i]) < 1e-12 * max(1.0, abs(w_abs[i])):    # (rounding of the two slopes)
            w_lin[i] = 0.0
        if w_abs[i] != 0.0:
            # Third pass: the kink to the LAST BIT.  With a < t < b close to it, cost(b) - cost(a) = w (a + b - 2 t) + v (b - a)
            # holds exactly in a, b (the doubles actually probed), so t follows with an error of a few ulp of w * delta --
            # far below half an ulp of t: rounding the result gives the double the environment holds (the pendulum's
            # target angle), where the crossing of two lines fitted at |x| ~ 2048 was only good to ~1e-15.
            delta = 2.0 ** -8 * max(1.0, abs(target[i]))
            a, b = float(target[i] - delta), float(target[i] + delta)
            x = np.zeros((2, n_s))
            x[:, i] = [a, b]
            ca, cb = hook(x)
            if np.isfinite(ca) and np.isfinite(cb):
                target[i] = 0.5 * (a + b) - ((cb - ca) - w_lin[i] * (b - a)) / (2.0 * w_abs[i])
    test = np.random.default_rng(0).normal(size=(16, n_s))
    want = (np.abs(target[None] - test) * w_abs[None]).sum(1) + test @ w_lin
    if np.allclose(hook(test), want, rtol=1e-9, atol=1e-9):
        return _lib.SX_OBJ_AFFINE_ABS, w_abs, target, w_lin
    return None


class CemSafeMPC(SafeMPC):
    """Safe MPC whose trajectory optimisation is the fused constrained CEM on the GPU."""

    def __init__(self, ssm: CemSSM, constraints, env, conf, opt_env, wx_feedback_cost, wu_feedback_cost,
                 beta_safety: float, safe_policy: Callable[[ndarray], ndarray],
                 lqr: Optional[LqrFeedbackController] = None, mpc=None) -> None:
        super().__init__()
        self._device = get_device(conf)
        self._env = env
        self._conf = conf
        self._state_dimen = env.n_s
        self._action_dimen = env.n_u
        self._l_mu = torch.tensor(env.l_mu, device=self._device)
        self._l_sigma = torch.tensor(env.l_sigm, device=self._device)
        self._get_random_action = env.random_action
        self._pq_flattener = PQFlattener(env.n_s)
        self._ssm = ssm
        self._constraints = constraints
        self._mpc_time_horizon = conf.mpc_time_horizon
        self._beta_safety = beta_safety
        self._safe_policy = safe_policy
        self._use_prior_model = conf.use_prior_model
        self._env_objective_cost_func = env.objective_cost_function
        self._record_rollouts = bool(getattr(conf, 'plot_cem_optimisation', False)
                                     or getattr(conf, 'plot_cem_terminal_states', False))

        linearized_model_a, linearized_model_b = opt_env['lin_model']
        self.lin_model = opt_env['lin_model']
        self._linearized_model_a = torch.tensor(linearized_model_a, device=self._device)
        self._linearized_model_b = torch.tensor(linearized_model_b, device=self._device)
        if lqr is None:
            lqr = LqrFeedbackController(wx_feedback_cost, wu_feedback_cost, env.n_s, env.n_u, linearized_model_a,
                                        linearized_model_b, conf=conf)
        self._lqr = lqr
        self._injected_mpc = mpc is not None
        self._mpc = mpc
        self._env_key = None
        self._objective_probe = torch.tensor([[0.3] * env.n_s, [-0.7] * env.n_s], dtype=torch.float64)
        self._last_mpc_actions = np.empty((0, self.action_dimen))
        self._mpc_actions_executed = 0
        self._batch_last_actions: Optional[List[ndarray]] = None   # per-episode ladder state of get_action_batch
        self._batch_executed: Optional[List[int]] = None
        self.last_rollouts: List[Rollouts] = []

    # ---- the reference's read-only members -----------------------------------------------------------------------
    @property
    def ssm(self) -> CemSSM:
        """exploration_runner saves ``safempc.ssm`` (reference exploration_runner.py:205-206)."""
        return self._ssm

    @property
    def state_dimen(self) -> int:
        return self._state_dimen

    @property
    def action_dimen(self) -> int:
        return self._action_dimen

    @property
    def safety_trajectory_length(self) -> int:
        return self._mpc_time_horizon

    @property
    def performance_trajectory_length(self) -> int:
        return 0  # no performance trajectory in the CEM solver (reference safempc_cem.py:212-215)

    @property
    def x_train(self) -> ndarray:
        x_train = self._ssm.x_train
        if x_train is None:
            return np.empty((0, self._state_dimen + self._action_dimen))
        return x_train.detach().cpu().numpy()

    def init_solver(self, cost_func=None) -> None:
        pass

    # ---- the fused optimiser -------------------------------------------------------------------------------------
    def _prior(self) -> Tuple[np.ndarray, np.ndarray]:
        a = self._linearized_model_a.cpu().numpy()
        b = self._linearized_model_b.cpu().numpy()
        if not self._use_prior_model:   # reference safempc_cem.py:291-296
            a, b = np.zeros_like(a), np.zeros_like(b)
        return a, b

    def _build_env(self) -> Tuple[_lib.SxEnv, bool]:
        """sx_env for the current problem; the bool says whether the objective must be evaluated through the hook."""
        a, b = self._prior()
        action_c = next(c for c in self._constraints if isinstance(c, ActionConstraint))
        state_c = next(c for c in self._constraints if hasattr(c, 'polytope_a'))
        spec = objective_spec(self._env)
        mode, w_abs, target, w_lin = spec if spec is not None else (_lib.SX_OBJ_NEG_VARIANCE, None, None, None)
        env = make_env(self._state_dimen, self._action_dimen, a=a, b=b, k_fb=self._lqr.get_control_matrix(),
                       l_mu=self._l_mu.cpu().numpy(), l_sigma=self._l_sigma.cpu().numpy(), beta=self._beta_safety,
                       h_mat=state_c.polytope_a, h_vec=state_c.polytope_b, u_min=action_c.u_min, u_max=action_c.u_max,
                       obj_mode=mode, obj_w_abs=w_abs, obj_target=target, obj_w_lin=w_lin, con_mode=state_c.mode)
        return env, spec is None

    def _solver(self):
        if self._injected_mpc:
            return self._mpc
        if getattr(self._ssm, 'kernel_family', None) not in ('rbf', 'feature', 'mlp', 'stepwise'):
            raise NotImplementedError('the fused CEM solver needs a HIP-backed CemSSM (GpCemSSM, McDropoutSSM, '
                                      'GalConcreteDropoutSSM, or JunkDimensionsSSM over one of them); other CemSSMs are outside '
                                      'the accelerated path')
        # the problem constants only change when the environment moves its objective (the pendulum's target angle,
        # environments.py:505-510): probe the hook at two fixed points and rebuild sx_env only when the answers change
        probe = self._env_objective_cost_func(self._objective_probe)
        key = None if probe is None else tuple(float(v) for v in probe.reshape(-1))
        if self._mpc is not None and key == self._env_key:
            return self._mpc
        env, needs_hook = self._build_env()
        if self._mpc is None:
            # cem_init_std: a scalar or one value per step; cem_warm_start: 'zero' (the reference's cold start) or
            # 'safe_policy' (FusedCemMpc.safe_policy_plan) -- both optional additions to the reference's config
            self._mpc = FusedCemMpc(self._ssm, env, self._mpc_time_horizon, self._conf.cem_num_rollouts,
                                    self._conf.cem_num_elites, self._conf.cem_num_iterations, device=self._device,
                                    seed=int(getattr(self._conf, 'cem_seed', 0)),
                                    init_std=getattr(self._conf, 'cem_init_std', 1.0),
                                    warm_start=getattr(self._conf, 'cem_warm_start', None) or 'zero',
                                    record_rollouts=self._record_rollouts)
        self._mpc.set_env(env, objective_hook=self._env_objective_cost_func if needs_hook else None)
        self._env_key = key
        return self._mpc

    def _flat_points(self, states: ndarray) -> Tensor:
        """Point states [E x n_s] as the optimiser's flat states [E x (n_s + n_s^2)] (PQFlattener.flatten(p, None): an
        all-zero Q block), assembled on the host: ONE host->device copy instead of a copy, a fill and a concatenation."""
        flat = np.zeros((states.shape[0], self._pq_flattener.get_flat_state_dimen()), dtype=np.float64)
        flat[:, :self._state_dimen] = states
        return torch.tensor(flat, device=self._device)

    def get_action(self, state: ndarray) -> Tuple[ndarray, MpcResult]:
        assert_shape(state, (self._state_dimen,))
        mpc_actions, rollouts = self._solver().get_actions(self._flat_points(np.asarray(state)[None]))
        mpc_actions = mpc_actions.detach().cpu().numpy() if mpc_actions is not None else mpc_actions
        self.last_rollouts = rollouts
        # the reference's ladder (safempc_cem.py:243-263): fresh solution, else the rest of the previous one, else the
        # safe controller
        if mpc_actions is not None:
            action = mpc_actions[0]
            self._last_mpc_actions = mpc_actions
            self._mpc_actions_executed = 1
            result = MpcResult.FOUND_SOLUTION
        elif self._mpc_actions_executed < self._last_mpc_actions.shape[0]:
            action = self._last_mpc_actions[self._mpc_actions_executed]
            self._mpc_actions_executed += 1
            result = MpcResult.PREVIOUS_SOLUTION
        else:
            action = self._safe_policy(state)
            result = MpcResult.SAFE_CONTROLLER
        return action, result

    def get_action_batch(self, states: ndarray, episode_ids=None, num_episodes: Optional[int] = None
                         ) -> Tuple[ndarray, List[MpcResult]]:
        """``get_action`` for independent episodes in lockstep (BASELINE config 5): states [A x n_s] ->
        (actions [A x n_u], one MpcResult per row).  One fused solve serves all rows; each episode keeps its own
        PREVIOUS_SOLUTION / SAFE_CONTROLLER ladder (reference safempc_cem.py:243-263).  `episode_ids` names the episode
        of each row (default 0 .. A-1) out of `num_episodes` -- a lockstep runner passes the episodes that are still
        running, and their ladders carry on where they were.  The ladder state is allocated by the first call and
        whenever `num_episodes` changes (``reset_batch`` starts over)."""
        states = np.asarray(states)
        if states.ndim != 2 or states.shape[1] != self._state_dimen:
            raise ValueError(f'Wanted shape (E, {self._state_dimen}), got {states.shape}')
        A = states.shape[0]
        ids = list(range(A)) if episode_ids is None else [int(e) for e in episode_ids]
        E = int(num_episodes) if num_episodes is not None else max(max(ids) + 1, A)
        if len(ids) != A or len(set(ids)) != A or min(ids) < 0 or max(ids) >= E:
            raise ValueError(f'episode_ids must name {A} distinct episodes out of {E}, got {ids}')
        if self._batch_last_actions is None or len(self._batch_last_actions) != E:
            self._batch_last_actions = [np.empty((0, self.action_dimen)) for _ in range(E)]
            self._batch_executed = [0] * E
        best, found, rollouts = self._solver().get_actions_batch(self._flat_points(states))
        best = best.detach().cpu().numpy()
        self.last_rollouts = rollouts
        actions: List[ndarray] = []
        results: List[MpcResult] = []
        for k, e in enumerate(ids):
            if bool(found[k]):
                self._batch_last_actions[e] = best[k]
                self._batch_executed[e] = 1
                actions.append(best[k][0])
                results.append(MpcResult.FOUND_SOLUTION)
            elif self._batch_executed[e] < self._batch_last_actions[e].shape[0]:
                actions.append(self._batch_last_actions[e][self._batch_executed[e]])
                self._batch_executed[e] += 1
                results.append(MpcResult.PREVIOUS_SOLUTION)
            else:
                actions.append(np.asarray(self._safe_policy(states[k])))
                results.append(MpcResult.SAFE_CONTROLLER)
        return np.stack(actions), results

    def reset_batch(self) -> None:
        self._batch_last_actions = None
        self._batch_executed = None

    def get_action_verbose(self, state: ndarray):
        raise NotImplementedError

    # ---- the reference's dynamics callback, kept for callers that step the model themselves ------------------------
    def _dynamics_func(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor]:
        """One particle-batch step on flat states (reference safempc_cem.py:288-302)."""
        ps, qs = self._pq_flattener.unflatten(states)
        a, b = self._prior()
        a, b = torch.tensor(a, device=states.device), torch.tensor(b, device=states.device)
        p_next, q_next, sigma = onestep_reachability(ps.contiguous(), self._ssm, actions, self._l_mu, self._l_sigma, qs,
                                                     k_fb=self._lqr.get_control_matrix_pytorch(), a=a, b=b, verbose=0,
                                                     c_safety=self._beta_safety)
        return self._pq_flattener.flatten(p_next, q_next), self._compute_objective_cost(p_next, sigma)

    def _compute_objective_cost(self, p_next: Tensor, sigma: Tensor) -> Tensor:
        objective_cost = self._env_objective_cost_func(p_next)
        if objective_cost is None:
            objective_cost = -torch.sum(sigma, dim=1)
        return objective_cost

    # ---- model maintenance ---------------------------------------------------------------------------------------
    def update_model(self, x: ndarray, y: ndarray, opt_hyp=False, replace_old=True, reinitialize_solver=True) -> None:
        """The model learns the error to the linear prior (reference safempc_cem.py:314-327)."""
        x_s, x_u = x[:, :self.state_dimen], x[:, self.state_dimen:]
        y_error = y - self.eval_prior(x_s, x_u) if self._use_prior_model else y
        self._ssm.update_model(torch.tensor(x, device=self._device), torch.tensor(y_error, device=self._device), opt_hyp,
                               replace_old)

    def information_gain(self) -> Union[ndarray, List[None]]:
        """Per-output information gain 1/2 log det(I + K / noise) when the model can report it (the reference returns
        a list of None, which exploration_runner.py:192 cannot store)."""
        if hasattr(self._ssm, 'information_gain'):
            return self._ssm.information_gain()
        return np.full(self.state_dimen, np.nan)

    def ssm_predict(self, z: ndarray) -> Tuple[ndarray, ndarray]:
        mean, sigma = self._ssm.predict_raw(torch.tensor(z, device=self._device))
        return mean.detach().cpu().numpy(), sigma.detach().cpu().numpy()

    def eval_prior(self, states: ndarray, actions: ndarray):
        a = self._linearized_model_a.cpu().numpy()
        b = self._linearized_model_b.cpu().numpy()
        return np.dot(states, a.T) + np.dot(actions, b.T)

    def collect_metrics(self) -> Dict[str, float]:
        return self._ssm.collect_metrics()
