"""Lockstep multi-episode driver around ``CemSafeMPC.get_action_batch`` (BASELINE config 5).

The reference steps ONE episode at a time: ``episode_runner.do_rollout`` (``safe_exploration/episode_runner.py:169-354``)
calls ``solver.get_action(state)`` once per environment step (:216-221), and the exploration module asks for one action
per iteration (``safempc_exploration.py:372-374``).  Here E independent episodes advance together: every step is ONE fused
solve for all episodes that are still running (the kernels carry the episode dimension), then one ``env.step`` per episode
on the host -- the simulators stay the reference's host code.  Per episode the function returns exactly what
``do_rollout`` returns and logs the same metrics, so ``run_episodic``'s bookkeeping (:100-139) can consume it unchanged.

Nothing here plots, renders or samples trajectories (the reference's optional diagnostics, out of scope).
"""
import time
import warnings
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .safempc_cem import MpcResult


@dataclass
class EpisodeResult:
    """One episode's ``do_rollout`` outputs: (xx, yy, cc, exit_codes, safety_failure) plus what it logs."""
    xx: np.ndarray                 # [T' x (n_s + n_u)] state-action pairs  (reference slicing ``[1:-1:obs_frequency]``)
    yy: np.ndarray                 # [T' x n_s]         observations
    cc: list                       # immediate costs (when a cost function is given)
    exit_codes: np.ndarray         # [T x 1]  1 = the MPC's own (or its previous) solution, 0 = safe controller, 5 = random
    safety_failure: bool           # the environment reported `done`
    episode_length: int = 0
    mpc_results: List[MpcResult] = field(default_factory=list)
    env_result: int = -1
    time_in_solver: float = 0.0

    def as_tuple(self) -> Tuple:
        return self.xx, self.yy, self.cc, self.exit_codes, self.safety_failure


def _exit_code(result: Optional[MpcResult]) -> int:
    if result is None:
        return 5                                                            # random action (reference :212)
    return 1 if result in (MpcResult.FOUND_SOLUTION, MpcResult.PREVIOUS_SOLUTION) else 0   # reference :223


def do_rollout_batch(envs: Sequence, n_steps: int, solver=None, metrics=None, episode_ids: Optional[Sequence[int]] = None,
                     cost: Optional[Callable] = None, mean=None, std=None, obs_frequency: int = 1,
                     verbosity: int = 0) -> List[EpisodeResult]:
    """E episodes in lockstep.  `envs`: one environment per episode (``reset(mean, std)``, ``step(action) -> (action,
    next_state, observation, done, env_result)``, ``random_action()``, ``collect_metrics()``, ``n_s``, ``n_u``).
    `solver`: a ``CemSafeMPC`` (``get_action_batch``) or None (random actions, exit code 5).  An episode whose
    environment reports `done` stops (safety failure, reference :311-313); the others carry on.
    """
    E = len(envs)
    ids = list(episode_ids) if episode_ids is not None else list(range(E))
    states = [np.asarray(env.reset(mean, std), dtype=np.float64) for env in envs]
    xx = [[np.zeros(envs[e].n_s + envs[e].n_u)] for e in range(E)]
    yy = [[np.zeros(envs[e].n_s)] for e in range(E)]
    exit_codes = [[0.0] for _ in range(E)]
    cc: List[list] = [[] for _ in range(E)]
    results: List[List[MpcResult]] = [[] for _ in range(E)]
    n_successful = [0] * E
    env_result = [-1] * E
    failed = [False] * E
    solver_time = [0.0] * E
    active = list(range(E))
    if solver is not None and hasattr(solver, 'reset_batch'):
        solver.reset_batch()
    for i in range(n_steps):
        if not active:
            break
        if solver is None:
            actions = [np.asarray(envs[e].random_action()) for e in active]
            step_results = [None] * len(active)
        else:
            t0 = time.time()
            batch = np.stack([states[e] for e in active])
            acts, step_results = solver.get_action_batch(batch, episode_ids=active, num_episodes=E)
            dt = time.time() - t0
            actions = [acts[k] for k in range(len(active))]
            for e in active:
                solver_time[e] += dt / len(active)      # the solve is shared: every episode is charged its share
            if verbosity > 0:
                print(f'total time solver in ms: {dt * 1e3:.3f} ({len(active)} episodes)')
        still = []
        for k, e in enumerate(active):
            action, next_state, observation, done, env_result[e] = envs[e].step(actions[k])
            if cost is not None:
                cc[e].append(cost(next_state))
            if step_results[k] is not None:
                results[e].append(step_results[k])
            xx[e].append(np.hstack((states[e], np.asarray(action).reshape(-1))))
            yy[e].append(np.asarray(observation).reshape(-1))
            exit_codes[e].append(float(_exit_code(step_results[k])))
            n_successful[e] += 1
            states[e] = np.asarray(next_state, dtype=np.float64)
            if done:
                failed[e] = True
            else:
                still.append(e)
        active = still
    out: List[EpisodeResult] = []
    for e in range(E):
        if metrics is not None:
            # the scalars do_rollout logs (reference :315-324)
            metrics.log_scalar('episode_length', n_successful[e], ids[e])
            metrics.log_scalar('mpc_found_solution_count', results[e].count(MpcResult.FOUND_SOLUTION), ids[e])
            metrics.log_scalar('mpc_previous_solution_count', results[e].count(MpcResult.PREVIOUS_SOLUTION), ids[e])
            metrics.log_scalar('safe_controller_fallback_count', results[e].count(MpcResult.SAFE_CONTROLLER), ids[e])
            metrics.log_scalar('mean_time_in_solver', float(solver_time[e]) / max(n_successful[e], 1), ids[e])
            metrics.log_scalar('env_result', env_result[e], ids[e])
            metrics.log_non_scalars(envs[e].collect_metrics(), ids[e])
            if solver is not None:
                metrics.log_non_scalars(solver.collect_metrics(), ids[e])
        if n_successful[e] == 0:
            warnings.warn('Agent survived 0 steps, cannot collect data')
            res = EpisodeResult(np.empty((0, envs[e].n_s + envs[e].n_u)), np.empty((0, envs[e].n_s)), [], np.empty((0, 1)),
                                failed[e])
        else:
            # the reference's slicing (:343-345): the leading zero row goes, and so does the LAST transition
            res = EpisodeResult(np.vstack(xx[e])[1:-1:obs_frequency, :], np.vstack(yy[e])[1:-1:obs_frequency, :], cc[e],
                                np.asarray(exit_codes[e])[1:, None], failed[e])
        res.episode_length, res.mpc_results, res.env_result, res.time_in_solver = \
            n_successful[e], results[e], env_result[e], solver_time[e]
        out.append(res)
    return out


def do_rollout(env, n_steps, scenario_id: int = 0, episode_id: int = 0, metrics=None, solver=None, cost=None,
               mean=None, std=None, obs_frequency: int = 1, verbosity: int = 0, **unused):
    """The reference's single-episode signature (episode_runner.py:169-177; plotting / rendering / sampling flags are
    accepted and ignored): one episode through the same loop."""
    return do_rollout_batch([env], n_steps, solver, metrics, [episode_id], cost, mean, std, obs_frequency,
                            verbosity)[0].as_tuple()
