"""Exploration module over a ready-made SafeMPC: the reference's ``DynamicSafeMPCExploration``
(``safe_exploration/safempc_exploration.py:357-393``), plus the multi-episode form of ``find_max_variance`` that
``exploration_runner`` can call once per iteration for all of its parallel explorations."""
from typing import List, Tuple

import numpy as np
from numpy import ndarray

from .safempc import SafeMPC


class DynamicSafeMPCExploration:
    def __init__(self, safempc: SafeMPC, env):
        self.safempc = safempc
        self.env = env
        self.n_s = safempc.state_dimen
        self.n_u = safempc.action_dimen
        self.n_safe = safempc.safety_trajectory_length
        self.n_perf = safempc.performance_trajectory_length
        self.safempc.init_solver(None)

    def find_max_variance(self, x_0: ndarray, sol_verbose: bool = False) -> Tuple[ndarray, ndarray]:
        """(x_0 [n_s x 1], u_apply [n_u x 1]) -- reference :372-374."""
        u_apply, _ = self.safempc.get_action(x_0)
        return x_0[:, None], u_apply[:, None]

    def find_max_variance_batch(self, x_0: ndarray) -> Tuple[ndarray, ndarray, List]:
        """E start states [E x n_s] -> (x_0 [E x n_s], u_apply [E x n_u], one MpcResult per episode): ONE fused solve."""
        u_apply, results = self.safempc.get_action_batch(np.atleast_2d(x_0))
        return np.atleast_2d(x_0), u_apply, results

    def find_max_variance_verbose(self, x_0: ndarray, sol_verbose: bool = False):
        return self.safempc.get_action_verbose(x_0)      # (raises NotImplementedError for the CEM solver, as the reference)

    def update_model(self, x, y, train=False, replace_old=False):
        self.safempc.update_model(x, y, train, replace_old)

    def get_information_gain(self):
        return self.safempc.information_gain()

    @property
    def x_train(self) -> ndarray:
        return self.safempc.x_train

    def ssm_predict(self, z: ndarray) -> Tuple[ndarray, ndarray]:
        return self.safempc.ssm_predict(z)
