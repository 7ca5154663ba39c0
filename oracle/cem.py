"""Oracle: CEM particle rollout, costs, ranking and elite refit (numpy float64).
TEST INFRASTRUCTURE - see oracle/__init__.py.

The optimiser loop itself is in the un-vendored ``constrained-cem-mpc`` submodule: PARITY UNPINNED.  What IS taken
from the reference (and pinned by its tests / call sites):

* dynamics callback: ``CemSafeMPC._dynamics_func``                       safe_exploration/safempc_cem.py:288-302
* objective: env hook or ``-sum(sigma)``                                 safempc_cem.py:304-312, environments.py:505-510
* state / terminal constraint cost ``10 x #not-inside``                  safempc_cem.py:102-132
* action constraint cost ``3`` per violating step                        test_safempc_cem.py:59-71
* ``get_actions`` returns ``(actions [H x n_u] | None, rollouts)``       safempc_cem.py:235, test_safempc_cem.py:83-148

The loop specification (DESIGN.md "CEM specification"): sample ``a = mean + std * eps`` per (t, action dim); roll every
particle out for H steps; order particles lexicographically by (constraint cost, objective cost, index); the first k
are the elites; refit mean / unbiased std from the elites; after the last iteration return the first-ranked action
sequence if its constraint cost is zero, else None.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import reachability as reach

ACTION_VIOLATION_COST = 3.0   # test_safempc_cem.py:59-71
STATE_VIOLATION_COST = 10.0   # safempc_cem.py:132

OBJ_NEG_VARIANCE = 0  # -sum_d sigma_d                           safempc_cem.py:308-311
OBJ_AFFINE_ABS = 1    # sum_j w_abs_j |c_j - p_j| + w_lin_j p_j   (pendulum environments.py:505-510, lander lunarlander.py:111-113)

CON_TERMINAL = 0      # EllipsoidTerminalConstraint: last state only      safempc_cem.py:102-113
CON_ALL_STATES = 1    # EllipsoidStateConstraint applied to every prefix  safempc_cem.py:116-132


@dataclass
class Problem:
    """Everything the rollout needs besides the GP: environment constants of SURVEY 8(d)."""
    n_s: int
    n_u: int
    a: np.ndarray            # [n_s x n_s] linear prior (zeros when use_prior_model is False, safempc_cem.py:291-296)
    b: np.ndarray            # [n_s x n_u]
    k_fb: np.ndarray         # [n_u x n_s]
    l_mu: np.ndarray         # [n_s]
    l_sigma: np.ndarray      # [n_s]
    beta: float              # c_safety
    h_mat: np.ndarray        # [m x n_s]
    h_vec: np.ndarray        # [m x 1]
    u_min: np.ndarray        # [n_u]
    u_max: np.ndarray        # [n_u]
    obj_mode: int = OBJ_NEG_VARIANCE
    obj_w_abs: Optional[np.ndarray] = None
    obj_target: Optional[np.ndarray] = None
    obj_w_lin: Optional[np.ndarray] = None
    con_mode: int = CON_ALL_STATES


@dataclass
class RolloutResult:
    traj_p: np.ndarray       # [P x H x n_s]
    traj_q: np.ndarray       # [P x H x n_s x n_s]
    sigma: np.ndarray        # [P x H x n_s]
    obj_cost: np.ndarray     # [P]
    con_cost: np.ndarray     # [P]
    status: int = 0


def objective_cost(prob: Problem, p_next, sigma):
    """safempc_cem.py:304-312."""
    if prob.obj_mode == OBJ_NEG_VARIANCE:
        return -sigma.sum(axis=1)
    w_abs = prob.obj_w_abs if prob.obj_w_abs is not None else np.zeros(prob.n_s)
    tgt = prob.obj_target if prob.obj_target is not None else np.zeros(prob.n_s)
    w_lin = prob.obj_w_lin if prob.obj_w_lin is not None else np.zeros(prob.n_s)
    return (np.abs(tgt[None] - p_next) * w_abs[None]).sum(1) + (p_next * w_lin[None]).sum(1)


def rollout(prob: Problem, gp, x0, actions, q0=None) -> RolloutResult:
    """H chained calls of the dynamics callback on the whole particle batch (safempc_cem.py:288-302).

    x0 [n_s] (or [P x n_s]); actions [P x H x n_u]; q0 None (point start, safempc_cem.py:234-235) or [n_s x n_s].
    """
    P, H, n_u = actions.shape
    n_s = prob.n_s
    p = np.broadcast_to(np.asarray(x0, dtype=np.float64).reshape(-1, n_s), (P, n_s)).copy()
    q = None if q0 is None else np.broadcast_to(np.asarray(q0, dtype=np.float64), (P, n_s, n_s)).copy()
    out = RolloutResult(np.empty((P, H, n_s)), np.empty((P, H, n_s, n_s)), np.empty((P, H, n_s)),
                        np.zeros(P), np.zeros(P))
    for t in range(H):
        u = actions[:, t, :]
        # PQFlattener round trip: all-zero Q over the whole batch is "None" (safempc_cem.py:69-71)
        if q is not None and np.count_nonzero(q) == 0:
            q = None
        p, q, sig, st = reach.onestep_reachability(p, gp, u, prob.l_mu, prob.l_sigma, q, prob.k_fb, prob.beta,
                                                   a=prob.a, b=prob.b)
        out.status |= st
        out.traj_p[:, t], out.traj_q[:, t], out.sigma[:, t] = p, q, sig
        out.obj_cost += objective_cost(prob, p, sig)
        viol_u = ((u < prob.u_min[None]) | (u > prob.u_max[None])).any(axis=1)
        out.con_cost += ACTION_VIOLATION_COST * viol_u
        if prob.con_mode == CON_ALL_STATES or t == H - 1:
            inside = reach.is_ellipsoid_inside_polytope(p, q, prob.h_mat, prob.h_vec)
            out.con_cost += STATE_VIOLATION_COST * (~inside)
    return out


def rank(con_cost, obj_cost, k):
    """Indices of the k best particles: lexicographic (constraint cost, objective cost, index), costs compared as numbers
    (-0.0 == +0.0); NaN sorts last, strictly behind +inf."""
    con_nan, obj_nan = np.isnan(con_cost), np.isnan(obj_cost)
    con = np.where(con_nan, np.inf, con_cost)
    obj = np.where(obj_nan, np.inf, obj_cost)
    order = np.lexsort((np.arange(len(con)), obj, obj_nan, con, con_nan))
    return order[:k]


def refit(elite_actions):
    """mean / unbiased std over the elite axis.  [k x H x n_u] -> ([H x n_u], [H x n_u]); std = 0 when k == 1."""
    k = elite_actions.shape[0]
    mean = elite_actions.mean(axis=0)
    if k > 1:
        std = np.sqrt(((elite_actions - mean[None]) ** 2).sum(axis=0) / (k - 1))
    else:
        std = np.zeros_like(mean)
    return mean, std


@dataclass
class CemTrace:
    means: list = field(default_factory=list)
    stds: list = field(default_factory=list)
    elites: list = field(default_factory=list)
    best_con: list = field(default_factory=list)


def cem_solve(prob: Problem, gp, x0, noise, num_elites, init_mean=None, init_std=None, q0=None):
    """The whole solve.  noise [iters x P x H x n_u] standard-normal draws (injected so both sides see the same).

    Returns (actions [H x n_u] or None, CemTrace).
    """
    iters, P, H, n_u = noise.shape
    mean = np.zeros((H, n_u)) if init_mean is None else np.array(init_mean, dtype=np.float64)
    std = np.ones((H, n_u)) if init_std is None else np.array(init_std, dtype=np.float64)
    trace = CemTrace()
    best = None
    for it in range(iters):
        actions = mean[None] + std[None] * noise[it]
        res = rollout(prob, gp, x0, actions, q0)
        idx = rank(res.con_cost, res.obj_cost, num_elites)
        mean, std = refit(actions[idx])
        trace.means.append(mean.copy())
        trace.stds.append(std.copy())
        trace.elites.append(idx.copy())
        trace.best_con.append(res.con_cost[idx[0]])
        best = actions[idx[0]].copy() if res.con_cost[idx[0]] == 0 else None
    return best, trace
