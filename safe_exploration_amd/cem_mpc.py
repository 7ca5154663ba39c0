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
from .gp_reachability_pytorch import raise_for_status
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
                status: Optional[Tensor] = None):
    """Thin wrapper over sx_cem_rollout.

    x0 [E x n_s]; either `actions` [E x P x H x n_u] (given) or (`mean`, `std` [E x H x n_u], `noise` [E x P x H x n_u]).
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
    model = ssm.device_model
    lib = _lib.lib()
    ws_bytes = int(lib.sx_cem_rollout_workspace_bytes(ctypes.byref(model), E, P, horizon))
    if ws_bytes < 0:
        raise _lib.SxError('sx_cem_rollout_workspace_bytes: bad arguments')
    workspace = ssm.workspace(ws_bytes)   # None on the fused path; cached on the model otherwise
    _lib.check(lib.sx_cem_rollout(ctypes.byref(model), ctypes.byref(env), E, P, horizon, _lib.ptr(x0.contiguous()),
                                  _lib.ptr(q0), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(noise), _lib.ptr(actions),
                                  _lib.ptr(traj), _lib.ptr(sigma), _lib.ptr(obj), _lib.ptr(con), _lib.ptr(status),
                                  _lib.ptr(workspace), ws_bytes, _lib.stream_ptr(dev)), 'sx_cem_rollout')
    return dict(actions=actions, obj_cost=obj, con_cost=con, traj=traj, sigma=sigma, status=status)


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


def fold_status(words) -> int:
    """Bitwise OR of the per-rank status words `solve` returns (host side: a handful of ints)."""
    out = 0
    for w in words.reshape(-1).tolist():
        out |= int(w)
    return out


class FusedCemMpc:
    """Drop-in for ``ConstrainedCemMpc``: ``get_actions(flat_state [1 x S]) -> (actions [H x n_u] | None, rollouts)``.

    With a process group of G > 1 ranks the particles are sharded (``num_rollouts`` is the GLOBAL count): every rank
    rolls out its share, keeps its local top-k rows, ONE all-reduce per iteration assembles the G*k candidates, and
    every rank redundantly ranks them and refits -- bit-identical on all ranks.
    """

    def __init__(self, ssm: GpCemSSM, env: _lib.SxEnv, time_horizon: int, num_rollouts: int, num_elites: int,
                 num_iterations: int, *, device=None, seed: int = 0, init_std: float = 1.0,
                 record_rollouts: bool = False, process_group=None):
        self._ssm = ssm
        self._env = env
        self._horizon = time_horizon
        self._num_iterations = num_iterations
        self._record = record_rollouts
        self._init_std = float(init_std)
        self._group = process_group
        self._world, self._rank = distributed.world_and_rank(process_group)
        self._num_rollouts = num_rollouts
        self._local_rollouts, _ = distributed.shard_particles(num_rollouts, self._world, self._rank)
        if num_elites > num_rollouts:
            raise ValueError(f'num_elites={num_elites} exceeds num_rollouts={num_rollouts}')
        self._num_elites = num_elites
        self._local_elites = min(num_elites, num_rollouts // self._world)  # same k on every rank
        self._device = torch.device(device if device is not None else 'cuda:0')
        self._gen = torch.Generator(device=self._device)
        self._gen.manual_seed(distributed.rank_seed(seed, self._rank))
        self.last_status = 0
        self._objective_hook = None
        # bench.py sets this to a list: (start, end) torch.cuda.Event pairs are then recorded around every
        # sx_cem_rollout launch, on the stream the kernel runs on
        self.rollout_events = None

    @property
    def num_iterations(self) -> int:
        return self._num_iterations

    def set_env(self, env: _lib.SxEnv, objective_hook=None) -> None:
        """New problem constants (the pendulum's objective target moves between calls).  `objective_hook`, if given, is
        an ``Environment.objective_cost_function`` this module has no kernel form for: it is then evaluated with torch
        on the recorded trajectory centres, H small launches per iteration instead of none."""
        self._env = env
        self._objective_hook = objective_hook

    def sample_noise(self, episodes: int = 1) -> Tensor:
        return torch.randn((episodes, self._local_rollouts, self._horizon, self._ssm.num_actions), dtype=torch.float64,
                           device=self._device, generator=self._gen)

    def solve(self, x0: Tensor, noise: Optional[Tensor] = None, init_mean: Optional[Tensor] = None,
              init_std: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, List[Rollouts], Tensor]:
        """E independent solves from x0 [E x n_s] (points).  Nothing here synchronises with the host.

        noise: optional [iters x E x P_local x H x n_u] pre-drawn standard normals (parity tests inject them).
        Returns (best [E x H x n_u], best_ok int32 [E], rollouts per iteration (if recorded), status int32 [G]: the
        status word of every rank, identical on all ranks; G = 1 without a process group -- OR them, `fold_status`).
        """
        n_u, H = self._ssm.num_actions, self._horizon
        E = x0.size(0)
        dev = x0.device
        L = H * n_u
        mean = torch.zeros((E, H, n_u), dtype=torch.float64, device=dev) if init_mean is None \
            else init_mean.to(dev).reshape(E, H, n_u).clone()
        std = torch.full((E, H, n_u), self._init_std, dtype=torch.float64, device=dev) if init_std is None \
            else init_std.to(dev).reshape(E, H, n_u).clone()
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        history: List[Rollouts] = []
        out = None
        xbuf = None
        if noise is None and 'sample_noise' not in vars(self):
            # one generator launch for the whole solve instead of one per iteration (a test that patches sample_noise
            # on the instance still gets its per-iteration calls)
            noise = torch.randn((self._num_iterations, E, self._local_rollouts, H, n_u), dtype=torch.float64,
                                device=self._device, generator=self._gen)
        for it in range(self._num_iterations):
            eps = noise[it] if noise is not None else self.sample_noise(E)
            if self.rollout_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(torch.cuda.current_stream(dev))
            r = cem_rollout(self._ssm, self._env, x0, H, mean=mean, std=std, noise=eps.contiguous(),
                            want_traj=self._record or self._objective_hook is not None, status=status)
            if self._objective_hook is not None:
                n_s = self._ssm.num_states
                centres = r['traj'][..., :n_s]                                   # [E x P x H x n_s]
                obj = torch.zeros_like(r['obj_cost'])
                for t in range(H):
                    obj += self._objective_hook(centres[:, :, t].reshape(-1, n_s)).reshape(obj.shape)
                r['obj_cost'] = obj.contiguous()
            if self.rollout_events is not None:
                ev[1].record(torch.cuda.current_stream(dev))
                self.rollout_events.append(ev)
            if self._world == 1:
                out = cem_rank_refit(r['con_cost'], r['obj_cost'], r['actions'], self._num_elites)
            else:
                k = self._local_elites
                G, n_slots = self._world, E * self._world * k * (2 + L)
                if xbuf is None:
                    # the zero-padded exchange buffers of ALL iterations in one allocation (one memset per solve); each
                    # carries G extra cells behind the slots: the status words ride along with the last exchange
                    xbuf = torch.zeros((self._num_iterations, n_slots + G), dtype=torch.float64, device=dev)
                slots = xbuf[it, :n_slots].view(E, G, k, 2 + L)
                if E == 1:
                    # the local elite rows go straight into this rank's slot: no copy between the kernel and the collective
                    cem_rank_refit(r['con_cost'], r['obj_cost'], r['actions'], k, want_refit=False,
                                   rows_out=slots[:, self._rank])
                else:
                    local = cem_rank_refit(r['con_cost'], r['obj_cost'], r['actions'], k, want_rows=True, want_refit=False)
                    slots[:, self._rank] = local['elite_rows']
                last = it == self._num_iterations - 1
                if last:
                    xbuf[it, n_slots + self._rank] = status[0]      # every rollout of this solve has been enqueued
                distributed.all_reduce_sum_(xbuf[it], self._group)               # the ONE collective of the iteration
                if last:
                    # every rank now holds the status words of all ranks: callers OR them, so all ranks raise together
                    status = xbuf[it, n_slots:].to(torch.int32)
                flat = slots.view(-1)
                out = cem_rank_refit(flat, flat[1:], flat[2:], self._num_elites, cost_stride=2 + L,
                                     act_stride=2 + L, row_len=L, num_candidates=G * k, num_problems=E)
            mean, std = out['mean'].view(E, H, n_u), out['std'].view(E, H, n_u)
            if self._record:
                for e in range(E):
                    history.append(Rollouts(r['traj'][e], r['actions'][e], r['obj_cost'][e], r['con_cost'][e]))
        return out['best'].view(E, H, n_u), out['best_ok'], history, status

    def get_actions_batch(self, states: Tensor) -> Tuple[Tensor, Tensor, List[Rollouts]]:
        """E independent episodes at once (BASELINE config 5): flat start states [E x (n_s + n_s^2)], all
        points.  One fused solve: the kernels carry the episode dimension, episodes never exchange anything.

        Returns (actions [E x H x n_u], found bool [E] on the host, rollouts); ``found[e] == False`` is the
        ``get_actions`` ``None`` of episode e.  Raises like ``get_actions`` if any episode hit a numerical failure.
        """
        n_s = self._ssm.num_states
        if states.dim() != 2 or states.size(1) != n_s + n_s * n_s:
            raise ValueError(f'Wanted shape (E, {n_s + n_s * n_s}), got {tuple(states.shape)}')
        if bool((states[:, n_s:] != 0).any()):
            raise NotImplementedError('get_actions_batch starts from point states (all-zero Q), as CemSafeMPC.get_action does')
        x0 = states[:, :n_s].to(self._device, torch.float64).contiguous()
        best, best_ok, history, status = self.solve(x0)
        flags = torch.cat((status, best_ok)).cpu()   # the one device->host hand-off of the batch
        self.last_status = fold_status(flags[:status.numel()])
        raise_for_status(self.last_status, 'get_actions_batch')
        return best, flags[status.numel():] != 0, history

    def get_actions(self, state: Tensor) -> Tuple[Optional[Tensor], List[Rollouts]]:
        """state: the flat start state [1 x (n_s + n_s^2)] with an all-zero Q block (a point, safempc_cem.py:234-235)."""
        n_s = self._ssm.num_states
        flat = state.reshape(1, -1)
        if flat.size(1) != n_s + n_s * n_s:
            raise ValueError(f'Wanted shape (1, {n_s + n_s * n_s}), got {tuple(state.shape)}')
        if bool((flat[:, n_s:] != 0).any()):
            raise NotImplementedError('get_actions starts from a point state (all-zero Q), as CemSafeMPC.get_action does')
        x0 = flat[:, :n_s].to(self._device, torch.float64).contiguous()
        best, best_ok, history, status = self.solve(x0)
        # the one device->host hand-off of a solve: status word + feasibility flag (+ the actions)
        flags = torch.cat((status, best_ok)).cpu()
        self.last_status = fold_status(flags[:status.numel()])
        raise_for_status(self.last_status, 'get_actions')
        if int(flags[status.numel()]) == 0:
            return None, history
        return best[0], history
