"""The solver interface the task runners program against (numpy in, numpy out).

Same members as the reference's ``safe_exploration/safempc.py:8-77``, so ``episode_runner`` / ``exploration_runner``
can drive either implementation.
"""
from abc import ABC, abstractmethod
from typing import Dict, Tuple

from numpy import ndarray


class SafeMPC(ABC):
    # ---- dimensions ----------------------------------------------------------------------------------------------
    @property
    @abstractmethod
    def state_dimen(self) -> int:
        ...

    @property
    @abstractmethod
    def action_dimen(self) -> int:
        ...

    @property
    @abstractmethod
    def safety_trajectory_length(self) -> int:
        ...

    @property
    @abstractmethod
    def performance_trajectory_length(self) -> int:
        ...

    # ---- the model -----------------------------------------------------------------------------------------------
    @property
    @abstractmethod
    def x_train(self) -> ndarray:
        """Inputs [N x (n_s + n_u)] the state-space model is currently conditioned on."""

    @abstractmethod
    def update_model(self, x: ndarray, y: ndarray, opt_hyp=False, replace_old=True, reinitialize_solver=True) -> None:
        ...

    @abstractmethod
    def ssm_predict(self, z: ndarray) -> Tuple[ndarray, ndarray]:
        ...

    @abstractmethod
    def eval_prior(self, states: ndarray, actions: ndarray):
        """Linear prior prediction [N x n_s] for states [N x n_s], actions [N x n_u]."""

    @abstractmethod
    def information_gain(self):
        ...

    # ---- solving -------------------------------------------------------------------------------------------------
    @abstractmethod
    def init_solver(self, cost_func=None) -> None:
        ...

    @abstractmethod
    def get_action(self, state: ndarray):
        ...

    @abstractmethod
    def get_action_verbose(self, state: ndarray):
        ...

    @abstractmethod
    def collect_metrics(self) -> Dict[str, float]:
        ...
