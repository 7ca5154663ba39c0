"""The constrained cross-entropy optimiser, fused onto the GPU.

Stands where the reference uses the third-party ``constrained_cem_mpc.ConstrainedCemMpc`` (call sites:
``safe_exploration/safempc_cem.py:7-8,139,193-196,235,270,278``).  That library drives the rollout through Python
callbacks -- H sequential ``DynamicsFunc`` calls and one ``Constraint`` call per trajectory per iteration.  Here one
CEM iteration is two launches: ``sx_cem_rollout`` (all particles x all H steps: sampling, GP predict, reachability,
objective and constraint costs) and ``sx_cem_rank_refit`` (ranking, elite refit, best feasible sequence).

The loop semantics are this repository's specification (DESIGN.md "CEM specification"; the library's source is not
available, "parity unpinned"); what the reference's tests pin -- ``get_actions`` returns ``(actions [H x n_u] | None,
rollouts)``, an action-constraint cost of 3 per violating step, 10 per state outside the polytope -- is kept.
"""
import ctypes
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib, distributed
from .gp_reachability_pytorch import raise_for_status, save_failure_state
from .ssm_cem.gp_ssm_cem import GpCemSSM


@dataclass
class Rollouts:
    """One CEM iteration's particles (what ``CemSafeMPC._plot_optimisation_process`` reads, safempc_cem.py:268-286)."""
    trajectories: Optional[Tensor]   # [P x H x (n_s + n_s^2)] flat states, or None when not recorded
    actions: Tensor                  # [P x H x n_u]
    objective_costs: Tensor          # [P]
    constraint_costs: Tensor         # [P]


def cem_rollout(ssm: GpCemSSM, env: _lib.SxEnv, x0: Tensor, horizon: int, *, actions: Optional[Tensor] = None,
                mean: Optional[Tensor] = None, std: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                q0: Optional[Tensor] = None, want_traj: bool = False, want_sigma: bool = False,
                status: Optional[Tensor] = None, elite_rows: Optional[Tensor] = None, want_dist: bool = False):
    """Thin wrapper over sx_cem_rollout / sx_cem_rollout_elites.

    x0 [E x n_s]; either `actions` [E x P x H x n_u] (given), or (`mean`, `std` [E x H x n_u], `noise` [E x P x H x n_u]),
    or (`elite_rows` [E x k x (2 + H n_u)], `noise`): the distribution is then refit from the previous iteration's elite
    rows in the rollout kernel's prologue (`fused_refit_applies`); `want_dist` returns that refit as `mean` / `std`.
    Returns dict(actions, obj_cost [E x P], con_cost [E x P], traj | None, sigma | None, status int32[1]).
    """
    n_s, n_u = ssm.num_states, ssm.num_actions
    _lib.require_gpu(x0, 'x0')
    dev = x0.device
    E = x0.size(0)
    if noise is not None:
        P = noise.size(1)
        actions = torch.empty((E, P, horizon, n_u), dtype=torch.float64, device=dev)
    else:
        P = actions.size(1)
        actions = actions.contiguous()
    S = n_s + n_s * n_s
    traj = torch.empty((E, P, horizon, S), dtype=torch.float64, device=dev) if want_traj else None
    sigma = torch.empty((E, P, horizon, n_s), dtype=torch.float64, device=dev) if want_sigma else None
    obj = torch.empty((E, P), dtype=torch.float64, device=dev)
    con = torch.empty((E, P), dtype=torch.float64, device=dev)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.lib()
    if getattr(ssm, 'kernel_family', 'rbf') == 'feature':
        # degenerate kernels ('linear', 'nn'): the weight-space rollout, one particle per lane (csrc/sx_feat.hpp)
        _lib.check(lib.sx_cem_rollout_feat(ctypes.byref(ssm.feat_model), ctypes.byref(env), E, P, horizon,
                                           _lib.ptr(x0.contiguous()), _lib.ptr(q0), _lib.ptr(mean), _lib.ptr(std),
                                           _lib.ptr(noise), _lib.ptr(actions), _lib.ptr(traj), _lib.ptr(sigma), _lib.ptr(obj),
                                           _lib.ptr(con), _lib.ptr(status), _lib.stream_ptr(dev)), 'sx_cem_rollout_feat')
        return dict(actions=actions, obj_cost=obj, con_cost=con, traj=traj, sigma=sigma, status=status)
    if getattr(ssm, 'kernel_family', 'rbf') == 'mlp':
        # MC-dropout ensembles over the frozen members: matrix cores for 1-2 hidden layers of <= 64 units
        # (csrc/sx_mlp_mfma.hpp), one particle per lane otherwise (csrc/sx_mlp.hpp)
        _lib.check(lib.sx_cem_rollout_mlp(ctypes.byref(ssm.mlp_model), ctypes.byref(env), E, P, horizon,
                                          _lib.ptr(x0.contiguous()), _lib.ptr(q0), _lib.ptr(mean), _lib.ptr(std),
                                          _lib.ptr(noise), _lib.ptr(actions), _lib.ptr(traj), _lib.ptr(sigma), _lib.ptr(obj),
                                          _lib.ptr(con), _lib.ptr(status), _lib.stream_ptr(dev)), 'sx_cem_rollout_mlp')
        return dict(actions=actions, obj_cost=obj, con_cost=con, traj=traj, sigma=sigma, status=status)
    model = ssm.device_model
    if elite_rows is not None:
        k = elite_rows.size(1)
        if noise is None or tuple(elite_rows.shape) != (E, k, 2 + horizon * n_u) or not elite_rows.is_contiguous():
            raise ValueError(f'elite_rows must be a contiguous [{E} x k x {2 + horizon * n_u}] tensor and come with noise')
        m_out = torch.empty((E, horizon, n_u), dtype=torch.float64, device=dev) if want_dist else None
        s_out = torch.empty((E, horizon, n_u), dtype=torch.float64, device=dev) if want_dist else None
        _lib.check(lib.sx_cem_rollout_elites(ctypes.byref(model), ctypes.byref(env), E, P, horizon, _lib.ptr(x0.contiguous()),
                                             _lib.ptr(q0), _lib.ptr(elite_rows), k, _lib.ptr(noise), _lib.ptr(actions),
                                             _lib.ptr(traj), _lib.ptr(sigma), _lib.ptr(obj), _lib.ptr(con), _lib.ptr(status),
                                             _lib.ptr(m_out), _lib.ptr(s_out), _lib.stream_ptr(dev)), 'sx_cem_rollout_elites')
        return dict(actions=actions, obj_cost=obj, con_cost=con, traj=traj, sigma=sigma, status=status, mean=m_out, std=s_out)
    ws_bytes = int(lib.sx_cem_rollout_workspace_bytes(ctypes.byref(model), E, P, horizon))
    if ws_bytes < 0:
        raise _lib.SxError('sx_cem_rollout_workspace_bytes: bad arguments')
    workspace = ssm.workspace(ws_bytes)   # None on the fused path; cached on the model otherwise
    _lib.check(lib.sx_cem_rollout(ctypes.byref(model), ctypes.byref(env), E, P, horizon, _lib.ptr(x0.contiguous()),
                                  _lib.ptr(q0), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(noise), _lib.ptr(actions),
                                  _lib.ptr(traj), _lib.ptr(sigma), _lib.ptr(obj), _lib.ptr(con), _lib.ptr(status),
                                  _lib.ptr(workspace), ws_bytes, _lib.stream_ptr(dev)), 'sx_cem_rollout')
    return dict(actions=actions, obj_cost=obj, con_cost=con, traj=traj, sigma=sigma, status=status)


def fused_refit_applies(ssm, episodes: int, particles: int, horizon: int, candidates: Optional[int] = None) -> bool:
    """May the elite refit move from the ranking kernel's tail into the next rollout's prologue (sx_cem_rollout_elites)?
    Exact-GP models on the single-launch path whose H n_u means and standard deviations fit the prologue's scratch, where
    the ranking that produces the rows (over `candidates` rows per problem; default: the particles) is the counting one."""
    if getattr(ssm, 'kernel_family', 'rbf') != 'rbf':
        return False
    if 2 * horizon * ssm.num_actions > 256 * (1 + ssm.num_states):
        return False
    lib = _lib.lib()
    if int(lib.sx_cem_rollout_workspace_bytes(ctypes.byref(ssm.device_model), episodes, particles, horizon)) != 0:
        return False
    # worth it where the ranking spreads over the chip and its refit would be a serial tail; many problems at once
    # (config 5: 8 episodes per GPU) rank and refit side by side, one workgroup each
    return int(lib.sx_cem_rank_counts(episodes, particles if candidates is None else candidates)) == 1


def cem_rollout_stepwise(ssm: GpCemSSM, env: _lib.SxEnv, x0: Tensor, actions: Tensor, *, status: Tensor, group=None,
                         objective_hook=None):
    """The rollout of ONE problem step by step, the way the reference's optimiser drives it: H dynamics-callback calls on
    the whole particle batch (safempc_cem.py:288-302), each = sx_gp_predict + sx_onestep_reach, then the costs.

    This is the path that keeps the reference's WHOLE-BATCH zero fix-up (gp_reachability_pytorch.py:234-243: an exact
    zero variance anywhere in the batch also lifts the negative ones), which the fused kernel cannot see across its
    workgroups.  `FusedCemMpc` falls back to it when a fused solve reports both SX_STATUS_NAN and SX_STATUS_ZERO_FIX --
    the only case in which the two rules can differ.  ~3 H launches per iteration instead of one; nothing synchronises.

    x0 [n_s]; actions [P x H x n_u] (this rank's particles); with a process group the batch spans the ranks: the
    "zero present" flag is all-reduced (MAX) per step.  Returns dict(obj_cost [P], con_cost [P]); `status` is OR-ed.
    """
    n_s, n_u = ssm.num_states, ssm.num_actions
    dev = actions.device
    P, H, _ = actions.shape
    lib = _lib.lib()
    arr = lambda field, r, c: torch.tensor(list(field)[:r * c], dtype=torch.float64, device=dev).view(r, c)
    u_min, u_max = arr(env.u_min, 1, n_u), arr(env.u_max, 1, n_u)
    w_abs, target, w_lin = arr(env.obj_w_abs, 1, n_s), arr(env.obj_target, 1, n_s), arr(env.obj_w_lin, 1, n_s)
    p = x0.reshape(1, n_s).expand(P, n_s).contiguous()
    q = None
    obj = torch.zeros(P, dtype=torch.float64, device=dev)
    con = torch.zeros(P, dtype=torch.float64, device=dev)
    d = torch.empty((P, env.m), dtype=torch.float64, device=dev)
    inside = torch.empty((P,), dtype=torch.uint8, device=dev)
    for t in range(H):
        u = actions[:, t].contiguous()
        if q is None:
            mean, var = ssm.predict_without_jacobians(p, u)
            jac = None
        else:
            mean, var, jac = ssm.predict_with_jacobians(p, u)
            jac = jac.contiguous()
        mean, var = mean.contiguous(), var.contiguous()     # (a wrapping CemSSM may hand back slices: JunkDimensionsSSM)
        if group is not None:
            # the batch spans the ranks: if ANY rank holds an exact zero, every rank lifts its non-positive variances
            zero_any = (var == 0).any().to(torch.int32).reshape(1)
            torch.distributed.all_reduce(zero_any, op=torch.distributed.ReduceOp.MAX, group=group)
            lifted = zero_any.bool() & (var <= 0)
            status |= torch.where(lifted.any(), _lib.SX_STATUS_ZERO_FIX, 0).to(torch.int32)
            var = torch.where(lifted, torch.full_like(var, 1e-5), var)
        p1, q1, sigma = torch.empty_like(p), torch.empty((P, n_s, n_s), dtype=torch.float64, device=dev), torch.empty_like(p)
        _lib.check(lib.sx_onestep_reach(ctypes.byref(env), P, _lib.ptr(p), _lib.ptr(q), _lib.ptr(u), _lib.ptr(mean),
                                        _lib.ptr(var.contiguous()), _lib.ptr(jac), _lib.ptr(p1), _lib.ptr(q1),
                                        _lib.ptr(sigma), _lib.ptr(status), _lib.stream_ptr(dev)), 'sx_onestep_reach')
        # costs: objective safempc_cem.py:304-312, action box test_safempc_cem.py:59-71, polytope safempc_cem.py:102-132
        if objective_hook is not None:
            obj = obj + objective_hook(p1)
        elif env.obj_mode == _lib.SX_OBJ_NEG_VARIANCE:
            obj = obj - sigma.sum(dim=1)
        else:
            obj = obj + (w_abs * (target - p1).abs() + w_lin * p1).sum(dim=1)
        con = con + _lib.SX_ACTION_VIOLATION_COST * ((u < u_min) | (u > u_max)).any(dim=1)
        if env.con_mode == _lib.SX_CON_ALL_STATES or t == H - 1:
            _lib.check(lib.sx_polytope_distance(ctypes.byref(env), P, _lib.ptr(p1), _lib.ptr(q1), 1.0, _lib.ptr(d),
                                                _lib.ptr(inside), _lib.stream_ptr(dev)), 'sx_polytope_distance')
            con = con + _lib.SX_STATE_VIOLATION_COST * (inside == 0)
        p, q = p1, q1
    return dict(obj_cost=obj, con_cost=con)


def cem_rank_refit(con: Tensor, obj: Tensor, actions: Tensor, k: int, *, cost_stride: int = 1,
                   act_stride: Optional[int] = None, row_len: Optional[int] = None, num_candidates: Optional[int] = None,
                   num_problems: Optional[int] = None, want_rows: bool = False, want_refit: bool = True,
                   rows_out: Optional[Tensor] = None):
    """Thin wrapper over sx_cem_rank_refit for E problems.

    Plain layout: con/obj [E x P], actions [E x P x ...].  Candidate-row layout (after the multi-GPU exchange): pass
    views into a [E x C x (2 + L)] buffer with cost_stride = act_stride = 2 + L, row_len = L, num_candidates = C,
    num_problems = E.  `rows_out` (contiguous [E x k x (2 + L)]) receives the elite rows in place of a fresh tensor: the
    multi-GPU solve hands in this rank's slot of the exchange buffer.
    """
    dev = con.device
    E = num_problems if num_problems is not None else con.size(0)
    P = num_candidates if num_candidates is not None else con.size(1)
    L = row_len if row_len is not None else actions[0, 0].numel()
    act_stride = act_stride if act_stride is not None else L
    idx = torch.empty((E, k), dtype=torch.int32, device=dev)
    rows = None
    if rows_out is not None:
        if tuple(rows_out.shape) != (E, k, 2 + L) or not rows_out.is_contiguous() or rows_out.dtype != torch.float64:
            raise ValueError(f'rows_out must be a contiguous float64 [{E} x {k} x {2 + L}] tensor')
        rows = rows_out
    elif want_rows:
        rows = torch.empty((E, k, 2 + L), dtype=torch.float64, device=dev)
    mean = torch.empty((E, L), dtype=torch.float64, device=dev) if want_refit else None
    std = torch.empty((E, L), dtype=torch.float64, device=dev) if want_refit else None
    best = torch.empty((E, L), dtype=torch.float64, device=dev)
    best_ok = torch.empty((E,), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().sx_cem_rank_refit(E, P, k, L, ctypes.c_void_p(con.data_ptr()), ctypes.c_void_p(obj.data_ptr()),
                                            cost_stride, ctypes.c_void_p(actions.data_ptr()), act_stride, _lib.ptr(idx),
                                            _lib.ptr(rows), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(best),
                                            _lib.ptr(best_ok), _lib.stream_ptr(dev)), 'sx_cem_rank_refit')
    return dict(elite_idx=idx, elite_rows=rows, mean=mean, std=std, best=best, best_ok=best_ok)


# limits of cem_rank_kernel (csrc/sx_rank.hpp: kRankThreads * kRankSlots candidates per problem, kRankMaxK elites)
RANK_MAX_CANDIDATES = 16384
RANK_MAX_ELITES = 2048


def rank_chunks(num_candidates: int) -> int:
    """Chunks a ranking of `num_candidates` is split into (1 = the kernel takes them at once): the smallest count that
    divides the candidates evenly into chunks of at most RANK_MAX_CANDIDATES."""
    c = -(-num_candidates // RANK_MAX_CANDIDATES)
    while num_candidates % c:
        c += 1
    return c


def cem_rank_refit_any(con: Tensor, obj: Tensor, actions: Tensor, k: int, **kw):
    """`cem_rank_refit` for any candidate count.  Beyond the kernel's 16 384 candidates per problem (BASELINE config 3's
    65 536 particles on ONE GPU) the ranking runs in two levels, exactly like the multi-GPU exchange without the collective:
    the C chunks of a problem hand in their top-k rows (one launch, E x C workgroups side by side), a second launch ranks
    the C k candidates.  `elite_idx` then numbers candidate rows, not particles (as after the exchange); equal
    (constraint, objective) pairs across chunks are ordered by chunk, not by particle index."""
    E, P = con.shape
    C = rank_chunks(P)
    if C == 1:
        return cem_rank_refit(con, obj, actions, k, **kw)
    Pc, L = P // C, actions[0, 0].numel()
    if k > Pc or C * k > RANK_MAX_CANDIDATES:
        raise ValueError(f'{P} candidates rank in {C} chunks of {Pc}: k={k} must not exceed the chunk size, nor '
                         f'{C} k the kernel\'s {RANK_MAX_CANDIDATES}')
    local = cem_rank_refit(con.reshape(E * C, Pc), obj.reshape(E * C, Pc), actions.reshape(E * C, Pc, L), k,
                           want_rows=True, want_refit=False)
    flat = local['elite_rows'].reshape(-1)                       # [E x C k x (2 + L)] candidate rows
    return cem_rank_refit(flat, flat[1:], flat[2:], k, cost_stride=2 + L, act_stride=2 + L, row_len=L,
                          num_candidates=C * k, num_problems=E, **kw)


def fold_status(words) -> int:
    """Bitwise OR of the per-rank status words `solve` returns (host side: a handful of ints)."""
    out = 0
    for w in words.reshape(-1).tolist():
        out |= int(w)
    return out


class FusedCemMpc:
    """Drop-in for ``ConstrainedCemMpc``: ``get_actions(flat_state [1 x S]) -> (actions [H x n_u] | None, rollouts)``.

    With a process group of G > 1 ranks the particles are sharded (``num_rollouts`` is the GLOBAL count): every rank
    rolls out its share, keeps its local top-k rows, ONE all-gather per iteration assembles the G*(k+1) candidate rows, and
    every rank redundantly ranks them and refits -- bit-identical on all ranks.
    """

    def __init__(self, ssm: GpCemSSM, env: _lib.SxEnv, time_horizon: int, num_rollouts: int, num_elites: int,
                 num_iterations: int, *, device=None, seed: int = 0, init_std=1.0, warm_start: str = 'zero',
                 record_rollouts: bool = False, process_group=None, force_exchange: bool = False):
        self._ssm = ssm
        self._env = env
        self._horizon = time_horizon
        self._num_iterations = num_iterations
        self._record = record_rollouts
        # the first iteration's sampling distribution: std is a scalar or one value per step [H] / [H x n_u];
        # warm_start 'zero' = mean 0 (the reference's cold start), 'safe_policy' = the safe controller u = k_fb x rolled
        # through the model's mean dynamics from x0 (safe_policy_plan)
        self._init_std = torch.as_tensor(init_std, dtype=torch.float64).cpu()
        if self._init_std.dim() > 0:
            self._init_std = self._init_std.reshape(time_horizon, -1).expand(time_horizon, ssm.num_actions).clone()
        if warm_start not in ('zero', 'safe_policy'):
            raise ValueError(f"warm_start must be 'zero' or 'safe_policy', got {warm_start!r}")
        self._warm_start = warm_start
        self._group = process_group
        self._world, self._rank = distributed.world_and_rank(process_group)
        # force_exchange: take the sharded code path (local ranking -> collective -> global ranking) even with ONE rank -- how
        # tests/test_gpu_distributed.py runs that path over RCCL (backend 'nccl') on a single-GPU box
        self._sharded = self._world > 1 or (force_exchange and process_group is not None)
        self._num_rollouts = num_rollouts
        self._local_rollouts, _ = distributed.shard_particles(num_rollouts, self._world, self._rank)
        if num_elites > num_rollouts:
            raise ValueError(f'num_elites={num_elites} exceeds num_rollouts={num_rollouts}')
        if num_elites > num_rollouts // self._world:
            # every rank hands in its local top-k rows; a rank that owns fewer than k particles could not, and the
            # global top-k would silently miss candidates
            raise ValueError(f'num_elites={num_elites} exceeds the smallest per-GPU share '
                             f'({num_rollouts // self._world} of {num_rollouts} particles over {self._world} GPUs)')
        self._num_elites = num_elites
        self._local_elites = num_elites     # the same k on every rank
        if num_elites > RANK_MAX_ELITES:
            raise ValueError(f'num_elites={num_elites} exceeds the ranking kernel\'s limit of {RANK_MAX_ELITES}')
        if self._sharded and self._world * (num_elites + 1) > RANK_MAX_CANDIDATES:
            raise ValueError(f'{self._world * (num_elites + 1)} candidate rows after the exchange exceed the ranking kernel\'s '
                             f'limit of {RANK_MAX_CANDIDATES}')
        chunks = rank_chunks(self._local_rollouts)     # > 1: two-level ranking on this GPU (cem_rank_refit_any)
        if chunks > 1 and (num_elites > self._local_rollouts // chunks or chunks * num_elites > RANK_MAX_CANDIDATES):
            raise ValueError(f'{self._local_rollouts} particles per GPU rank in {chunks} chunks: num_elites={num_elites} is '
                             f'too large for it (limit {min(self._local_rollouts // chunks, RANK_MAX_CANDIDATES // chunks)})')
        self._device = torch.device(device if device is not None else 'cuda:0')
        self._init_std = self._init_std.to(self._device)
        # the iteration's collective on the compute stream (our own RCCL communicator) where the group is an nccl one
        self._comm = distributed.make_comm(process_group, self._device) if self._sharded else None
        self._gen = torch.Generator(device=self._device)
        self._gen.manual_seed(distributed.rank_seed(seed, self._rank))
        self.last_status = 0
        # a CemSSM that only offers the predict_* surface (JunkDimensionsSSM over a HIP-backed model) is always rolled out
        # step by step: sx_gp_predict through the wrapper + sx_onestep_reach per step, the reference's own call pattern
        self._always_stepwise = getattr(ssm, 'kernel_family', None) == 'stepwise'
        self.stepwise_fallbacks = 0     # solves repeated through the step-by-step path (see _solve_checked)
        self._last_noise = self._last_actions = None
        self._objective_hook = None
        # bench.py sets this to a list: (start, end) torch.cuda.Event pairs are then recorded around every
        # sx_cem_rollout launch, on the stream the kernel runs on
        self.rollout_events = None
        # likewise around the multi-GPU part of an iteration (collective + global ranking launch): bench.py's exchange_us.
        # Every `exchange_event_stride`-th exchange only: an event pair around EVERY exchange cost 7 us per iteration
        # (measured with one RCCL rank), 4 % of a sharded config-2 iteration
        self.exchange_events = None
        self.exchange_event_stride = 16
        self._exchanges_seen = 0

    @property
    def num_iterations(self) -> int:
        return self._num_iterations

    def set_env(self, env: _lib.SxEnv, objective_hook=None) -> None:
        """New problem constants (the pendulum's objective target moves between calls).  `objective_hook`, if given, is
        an ``Environment.objective_cost_function`` this module has no kernel form for: it is then evaluated with torch
        on the recorded trajectory centres, H small launches per iteration instead of none."""
        self._env = env
        self._objective_hook = objective_hook
        self._prior_tensors = None

    def _prior(self):
        """(a [n_s x n_s], b [n_s x n_u], k_fb [n_u x n_s]) of the current sx_env as device tensors."""
        if getattr(self, '_prior_tensors', None) is None:
            n_s, n_u = self._ssm.num_states, self._ssm.num_actions
            t = lambda arr, r, c: torch.tensor(list(arr)[:r * c], dtype=torch.float64, device=self._device).view(r, c)
            self._prior_tensors = (t(self._env.a, n_s, n_s), t(self._env.b, n_s, n_u), t(self._env.k_fb, n_u, n_s))
        return self._prior_tensors

    def safe_policy_plan(self, x0: Tensor) -> Tensor:
        """[E x H x n_u]: the safe controller u_t = k_fb x_t (reference safempc_cem.py:259-262, the last rung of the
        fallback ladder) rolled through the model's MEAN dynamics x_{t+1} = a x_t + b u_t + mu(x_t, u_t) from x0
        [E x n_s] -- H one-point-per-episode sx_gp_predict launches, nothing synchronises.  It is the warm start of
        workloads whose open-loop instability (cart-pole: 1.77 per step, 9e4 over H = 20) leaves a zero-mean start no
        feasible particle to learn from."""
        a, b, k_fb = self._prior()
        x = x0.to(self._device, torch.float64)
        plan = []
        for _ in range(self._horizon):
            u = x @ k_fb.t()
            mean, _ = self._ssm.predict_without_jacobians(x.contiguous(), u.contiguous())
            plan.append(u)
            x = x @ a.t() + u @ b.t() + mean
        return torch.stack(plan, dim=1)

    def _constant(self, name: str, episodes: int, dev, make) -> Tensor:
        """A read-only device tensor that depends on (name, episodes, device) only, built once."""
        cache = self.__dict__.setdefault('_constants', {})
        key = (name, episodes, str(dev))
        if key not in cache:
            cache[key] = make()
        return cache[key]

    def _fresh_status(self, dev) -> Tensor:
        """A zeroed int32[1] status word: slices of a pool that is zeroed once per 256 solves (one fill launch instead of
        256).  A slice is handed out once per pool generation; callers read it right after the solve."""
        pool = getattr(self, '_status_pool', None)
        if pool is None or self._status_next >= pool.numel() or pool.device != dev:
            self._status_pool = pool = torch.zeros(256, dtype=torch.int32, device=dev)
            self._status_next = 0
        i = self._status_next
        self._status_next += 1
        return pool[i:i + 1]

    def _exchange(self, episodes: int, k: int, row_len: int, dev):
        """The exchange buffers of a solve.  One problem (all-gather mode): every cell is overwritten by each solve and the
        padding rows never change, so the object is built once and kept -- four small launches per solve otherwise.  Several
        problems (all-reduce over zero padding): fresh, zeroed buffers per solve."""
        if episodes != 1:
            return distributed.EliteExchange(self._num_iterations, episodes, k, row_len, self._group, dev, comm=self._comm)
        key = (k, row_len, str(dev))
        cache = self.__dict__.setdefault('_xch_cache', {})
        if key not in cache:
            cache[key] = distributed.EliteExchange(self._num_iterations, 1, k, row_len, self._group, dev, comm=self._comm)
        return cache[key]

    def _next_noise(self, episodes: int) -> Tensor:
        """[iters x E x P_local x H x n_u] standard normals for one solve, from this solver's generator.  They are drawn for
        up to 8 solves per generator launch (at most 256 MB): one launch per solve is 8 us + a 6 us gap in front of the first
        rollout, 1.2 % of a config-2 solve."""
        shape = (self._num_iterations, episodes, self._local_rollouts, self._horizon, self._ssm.num_actions)
        pool = getattr(self, '_noise_pool', None)
        if pool is None or tuple(pool.shape[1:]) != shape or self._noise_next >= pool.size(0):
            per_solve = 8
            for n in shape:
                per_solve *= n
            batch = max(1, min(8, (256 << 20) // max(per_solve, 1)))
            self._noise_pool = pool = torch.randn((batch,) + shape, dtype=torch.float64, device=self._device,
                                                  generator=self._gen)
            self._noise_next = 0
        out = pool[self._noise_next]
        self._noise_next += 1
        return out

    def sample_noise(self, episodes: int = 1) -> Tensor:
        return torch.randn((episodes, self._local_rollouts, self._horizon, self._ssm.num_actions), dtype=torch.float64,
                           device=self._device, generator=self._gen)

    def solve(self, x0: Tensor, noise: Optional[Tensor] = None, init_mean: Optional[Tensor] = None,
              init_std: Optional[Tensor] = None, stepwise: bool = False) -> Tuple[Tensor, Tensor, List[Rollouts], Tensor]:
        """E independent solves from x0 [E x n_s] (points).  Nothing here synchronises with the host.

        stepwise: roll out through `cem_rollout_stepwise` (H x (sx_gp_predict + sx_onestep_reach) per iteration, the
        reference's whole-batch zero fix-up) instead of the fused kernel; `get_actions` uses it to settle the one case
        in which the fused kernel's per-particle fix-up can differ from the reference's.

        noise: optional [iters x E x P_local x H x n_u] pre-drawn standard normals (parity tests inject them).
        Returns (best [E x H x n_u], best_ok int32 [E], rollouts per iteration (if recorded), status int32 [G]: the
        status word of every rank, identical on all ranks; G = 1 without a process group -- OR them, `fold_status`).
        """
        n_u, H = self._ssm.num_actions, self._horizon
        E = x0.size(0)
        dev = x0.device
        L = H * n_u
        # (the constant start distribution and the zeroed status words are kept between solves: three small launches per
        # solve otherwise, 2 % of a config-2 solve)
        if init_mean is not None:
            mean = init_mean.to(dev).reshape(E, H, n_u).clone()
        elif self._warm_start == 'safe_policy':
            mean = self.safe_policy_plan(x0).contiguous()
        else:
            mean = self._constant('zero_mean', E, dev, lambda: torch.zeros((E, H, n_u), dtype=torch.float64, device=dev))
        if init_std is not None:
            std = init_std.to(dev).reshape(E, H, n_u).clone()
        else:
            std = self._constant('init_std', E, dev, lambda: self._init_std.to(dev).expand(E, H, n_u).contiguous())
        status = self._fresh_status(dev)
        history: List[Rollouts] = []
        out = None
        xch = None
        stepwise = stepwise or self._always_stepwise
        if noise is None and 'sample_noise' not in vars(self):
            # one generator launch for the whole solve instead of one per iteration (a test that patches sample_noise
            # on the instance still gets its per-iteration calls)
            noise = self._next_noise(E)
        self._last_noise, self._last_actions = noise, None
        # From the second iteration on the refit happens in the rollout kernel's prologue, straight from the elite rows of
        # the ranking before it (sx_cem_rollout_elites): the ranking launches then skip their refit tail.
        chunks = rank_chunks(self._local_rollouts)
        final_candidates = (self._world * (self._local_elites + (1 if E == 1 else 0)) if self._sharded
                            else chunks * self._num_elites if chunks > 1 else self._local_rollouts)
        in_prologue = (not stepwise) and fused_refit_applies(self._ssm, E, self._local_rollouts, H, final_candidates)
        rows = None
        for it in range(self._num_iterations):
            eps = noise[it] if noise is not None else self.sample_noise(E)
            if noise is None:
                self._last_noise = None      # per-iteration draws (a patched sample_noise): nothing to replay
            if self.rollout_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(torch.cuda.current_stream(dev))
            if stepwise:
                acts = (mean.unsqueeze(1) + std.unsqueeze(1) * eps).contiguous()       # [E x P x H x n_u]
                per_e = [cem_rollout_stepwise(self._ssm, self._env, x0[e], acts[e], status=status,
                                              group=self._group if self._world > 1 else None,
                                              objective_hook=self._objective_hook) for e in range(E)]
                r = dict(actions=acts, traj=None, obj_cost=torch.stack([q['obj_cost'] for q in per_e]),
                         con_cost=torch.stack([q['con_cost'] for q in per_e]))
            elif rows is not None:
                r = cem_rollout(self._ssm, self._env, x0, H, elite_rows=rows, noise=eps.contiguous(),
                                want_traj=self._record or self._objective_hook is not None, status=status)
            else:
                r = cem_rollout(self._ssm, self._env, x0, H, mean=mean, std=std, noise=eps.contiguous(),
                                want_traj=self._record or self._objective_hook is not None, status=status)
            if self._objective_hook is not None and not stepwise:
                n_s = self._ssm.num_states
                centres = r['traj'][..., :n_s]                                   # [E x P x H x n_s]
                obj = torch.zeros_like(r['obj_cost'])
                for t in range(H):
                    obj += self._objective_hook(centres[:, :, t].reshape(-1, n_s)).reshape(obj.shape)
                r['obj_cost'] = obj.contiguous()
            if self.rollout_events is not None:
                ev[1].record(torch.cuda.current_stream(dev))
                self.rollout_events.append(ev)
            if not self._sharded:
                out = cem_rank_refit_any(r['con_cost'], r['obj_cost'], r['actions'], self._num_elites,
                                         want_rows=in_prologue, want_refit=not in_prologue)
            else:
                k = self._local_elites
                if xch is None:
                    xch = self._exchange(E, k, L, dev)
                timed = self.exchange_events is not None and self._exchanges_seen % self.exchange_event_stride == 0
                self._exchanges_seen += 1
                if timed:   # four marks: before the local ranking | before the collective | after it | after the global ranking
                    xev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    xev[0].record(torch.cuda.current_stream(dev))
                if E == 1:
                    # the local elite rows go straight into this rank's slot: no copy between the kernel and the collective
                    cem_rank_refit_any(r['con_cost'], r['obj_cost'], r['actions'], k, want_refit=False,
                                       rows_out=xch.local_slot(it))
                else:
                    local = cem_rank_refit_any(r['con_cost'], r['obj_cost'], r['actions'], k, want_rows=True, want_refit=False)
                    xch.local_slot(it).copy_(local['elite_rows'])
                last = it == self._num_iterations - 1
                if timed:
                    xev[1].record(torch.cuda.current_stream(dev))
                # the ONE collective of the iteration; on the last one the status words of all ranks ride along
                # (every rollout of this solve has been enqueued by then)
                cand, words = xch.exchange(it, status if last else None)
                if timed:
                    xev[2].record(torch.cuda.current_stream(dev))
                if last:
                    status = words
                flat = cand.reshape(-1)
                out = cem_rank_refit(flat, flat[1:], flat[2:], self._num_elites, cost_stride=2 + L,
                                     act_stride=2 + L, row_len=L, num_candidates=xch.candidates, num_problems=E,
                                     want_rows=in_prologue, want_refit=not in_prologue)
                if timed:
                    xev[3].record(torch.cuda.current_stream(dev))
                    self.exchange_events.append(xev)
            if in_prologue:
                rows = out['elite_rows']
            else:
                mean, std = out['mean'].view(E, H, n_u), out['std'].view(E, H, n_u)
            self._last_actions = r['actions']
            if self._record and r['traj'] is not None:
                for e in range(E):
                    history.append(Rollouts(r['traj'][e], r['actions'][e], r['obj_cost'][e], r['con_cost'][e]))
        return out['best'].view(E, H, n_u), out['best_ok'], history, status

    def _solve_checked(self, x0: Tensor, where: str, q_block: Optional[Tensor] = None):
        """`solve` + the ONE device->host hand-off of a solve + the reference's failure behaviour.  The hand-off carries the
        status words, the feasibility flags, the selected actions and "is any entry of `q_block` non-zero" (the callers'
        point-state check, evaluated on the device and read here instead of in a synchronisation of its own before the
        solve).  Returns (best [E x H x n_u] ON THE HOST, found bool [E] on the host, rollouts)."""
        if q_block is not None and not q_block.is_contiguous():
            q_block = q_block.contiguous()

        def hand_off(best, best_ok, status):
            # one launch packs [status words | flags | point-state check | actions], one copy into pinned memory brings
            # them over (sx_cem_pack_result)
            G, E, L = status.numel(), best_ok.numel(), best[0].numel()
            n = G + E + 1 + E * L
            packed = torch.empty(n, dtype=torch.float64, device=best.device)
            _lib.check(_lib.lib().sx_cem_pack_result(G, E, L, _lib.ptr(status), _lib.ptr(best_ok), _lib.ptr(q_block),
                                                     q_block.numel() if q_block is not None else 0,
                                                     _lib.ptr(best.contiguous()), _lib.ptr(packed),
                                                     _lib.stream_ptr(best.device)), 'sx_cem_pack_result')
            host = getattr(self, '_pinned', None)
            if host is None or host.numel() < n:
                self._pinned = host = torch.empty(max(n, 256), dtype=torch.float64).pin_memory()
            host[:n].copy_(packed, non_blocking=True)
            torch.cuda.current_stream(best.device).synchronize()
            out = host[:n].clone()
            return (out[:G].to(torch.int64), out[G:G + E] != 0, bool(out[G + E] != 0),
                    out[G + E + 1:].view(best.shape))

        best, best_ok, history, status = self.solve(x0)
        words, found, is_nonpoint, best_host = hand_off(best, best_ok, status)
        if is_nonpoint:
            raise NotImplementedError(f'{where} starts from point states (all-zero Q), as CemSafeMPC.get_action does')
        self.last_status = fold_status(words)
        both = _lib.SX_STATUS_NAN | _lib.SX_STATUS_ZERO_FIX
        if (self.last_status & both) == both and self._last_noise is not None:
            # The fused kernel lifts exact-zero variances per particle; the reference decides on the whole batch: with a
            # zero present it lifts the NEGATIVE variances too and carries on, where the kernel went to sqrt -> NaN
            # (gp_reachability_pytorch.py:234-243).  Both bits set is the only case in which that can matter: repeat
            # the solve with the same draws through the step-by-step path, which follows the reference's rule.
            self.stepwise_fallbacks += 1
            best, best_ok, history, status = self.solve(x0, noise=self._last_noise, stepwise=True)
            words, found, _, best_host = hand_off(best, best_ok, status)
            self.last_status = fold_status(words)
        raise_for_status(self.last_status, where,
                         dump=lambda: save_failure_state(self._ssm, x0, self._last_actions))
        return best_host, found, history

    def get_actions_batch(self, states: Tensor) -> Tuple[Tensor, Tensor, List[Rollouts]]:
        """E independent episodes at once (BASELINE config 5): flat start states [E x (n_s + n_s^2)], all
        points.  One fused solve: the kernels carry the episode dimension, episodes never exchange anything.

        Returns (actions [E x H x n_u] on the host, found bool [E] on the host, rollouts); ``found[e] == False`` is the
        ``get_actions`` ``None`` of episode e.  Raises like ``get_actions`` if any episode hit a numerical failure.
        """
        n_s = self._ssm.num_states
        if states.dim() != 2 or states.size(1) != n_s + n_s * n_s:
            raise ValueError(f'Wanted shape (E, {n_s + n_s * n_s}), got {tuple(states.shape)}')
        states = states.to(self._device, torch.float64)
        return self._solve_checked(states[:, :n_s].contiguous(), 'get_actions_batch', q_block=states[:, n_s:])

    def get_actions(self, state: Tensor) -> Tuple[Optional[Tensor], List[Rollouts]]:
        """state: the flat start state [1 x (n_s + n_s^2)] with an all-zero Q block (a point, safempc_cem.py:234-235).
        The selected actions come back on the host (they travel with the solve's one device->host hand-off)."""
        n_s = self._ssm.num_states
        flat = state.reshape(1, -1)
        if flat.size(1) != n_s + n_s * n_s:
            raise ValueError(f'Wanted shape (1, {n_s + n_s * n_s}), got {tuple(state.shape)}')
        flat = flat.to(self._device, torch.float64)
        best, found, history = self._solve_checked(flat[:, :n_s].contiguous(), 'get_actions', q_block=flat[:, n_s:])
        if not bool(found[0]):
            return None, history
        return best[0], history
