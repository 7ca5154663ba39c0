"""State-space-model interface consumed by the CEM safe-MPC solver (torch tensors in, torch tensors out).

Mirrors the surface of the reference's ``safe_exploration/ssm_cem/ssm_cem.py:11-131`` so a model written against one
works with the other.
"""
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from ..utils import assert_shape


class CemSSM(ABC):
    def __init__(self, state_dimen: int, action_dimen: int):
        self.num_states = state_dimen
        self.num_actions = action_dimen
        self._x_train: Optional[Tensor] = None
        self._y_train: Optional[Tensor] = None

    # ---- training data -------------------------------------------------------------------------------------------
    @property
    def x_train(self) -> Optional[Tensor]:
        return self._x_train

    @property
    def y_train(self) -> Optional[Tensor]:
        return self._y_train

    def update_model(self, train_x: Tensor, train_y: Tensor, opt_hyp=False, replace_old=False) -> None:
        """train_x [N x (n_s+n_u)], train_y [N x n_s]; merges with the stored data unless replace_old.

        Hyper-parameters are re-fitted when opt_hyp is set or the model is parametric.
        """
        n = train_x.size(0)
        assert_shape(train_x, (n, self.num_states + self.num_actions))
        assert_shape(train_y, (n, self.num_states))
        if not replace_old and self._x_train is not None and self._y_train is not None:
            train_x = torch.cat((self._x_train, train_x), dim=0)
            train_y = torch.cat((self._y_train, train_y), dim=0)
        self._x_train, self._y_train = train_x, train_y
        self._update_model(train_x, train_y)
        if opt_hyp or self.parametric:
            self._train_model(train_x, train_y)

    @abstractmethod
    def _update_model(self, x_train: Tensor, y_train: Tensor) -> None:
        ...

    @abstractmethod
    def _train_model(self, x_train: Tensor, y_train: Tensor) -> None:
        ...

    # ---- prediction ----------------------------------------------------------------------------------------------
    @abstractmethod
    def predict_with_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """states [N x n_s], actions [N x n_u] -> mean [N x n_s], var [N x n_s], jac [N x n_s x (n_s+n_u)]."""

    @abstractmethod
    def predict_without_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor]:
        """states [N x n_s], actions [N x n_u] -> mean [N x n_s], var [N x n_s]."""

    @abstractmethod
    def predict_raw(self, z: Tensor) -> Tuple[Tensor, Tensor]:
        """z [N x (n_s+n_u)] -> mean [n_s x N], var [n_s x N] (outputs first, as the reference's GpCemSSM)."""

    def _join_states_actions(self, states: Tensor, actions: Tensor) -> Tensor:
        n = states.size(0)
        assert_shape(states, (n, self.num_states))
        assert_shape(actions, (n, self.num_actions))
        return torch.cat((states, actions), dim=1)

    # ---- bookkeeping ---------------------------------------------------------------------------------------------
    @abstractmethod
    def collect_metrics(self) -> Dict[str, Any]:
        ...

    @property
    @abstractmethod
    def parametric(self) -> bool:
        ...


class JunkDimensionsSSM(CemSSM):
    """Wraps an SSM, padding every call with zero-valued junk state / action dimensions (the reference's way of trying
    an SSM in a higher-dimensional setting, ssm_cem/ssm_cem.py:134-210; created by utils_config._create_cem_ssm:46-47).

    Mirrors the reference exactly, including where the junk goes: states -> [states, junk], actions -> [actions, junk],
    raw inputs -> [z, all junk]; outputs are cut back to the leading real dimensions (the Jacobian to its leading
    n_s + n_u columns, as the reference does).  It works with any CemSSM at this surface -- the HIP-backed GpCemSSM
    directly while the padded sizes stay within its limits (n_s <= 4, n_u <= 2), and beyond them in a FOLDED form that gives
    the padded model's numbers (`_construct_folded`).

    The CEM solver (CemSafeMPC / FusedCemMpc) takes the wrapper too, through its STEP-BY-STEP rollout (`kernel_family =
    'stepwise'`: H x (predict through this wrapper + sx_onestep_reach) per CEM iteration, the way the reference's optimiser
    drives its dynamics callback), not through the fused kernel: the wrapper's placement of the junk is not a GP over the
    real dimensions -- training rows are [z, junk] but queries [states, junk, actions, junk], so with junk states the
    action meets training columns that only ever held zeros, and the "action" columns of the returned Jacobian are
    derivatives with respect to junk STATE inputs -- and the step-by-step path reproduces exactly that, whatever it means.
    """
    kernel_family = 'stepwise'

    def __init__(self, constructor: Callable[..., CemSSM], state_dimen: int, action_dimen: int, junk_states: int,
                 junk_actions: int):
        super().__init__(state_dimen, action_dimen)
        self._junk_states = junk_states
        self._junk_actions = junk_actions
        self._cols: Optional[Tensor] = None      # folded form: the padded input columns the inner model keeps
        try:
            self._ssm = constructor(state_dimen=state_dimen + junk_states, action_dimen=action_dimen + junk_actions)
        except ValueError as too_large:
            self._ssm = self._construct_folded(constructor, too_large)

    def _construct_folded(self, constructor: Callable[..., CemSSM], too_large: Exception) -> CemSSM:
        """Padded sizes beyond the inner model's limits (the reference's experiment goes to 5 junk states,
        notebooks/results.ipynb cell 15: 7 states for the pendulum).  For an exact GP with an ARD RBF kernel the padded model
        FOLDS exactly: an input column that is zero in every training row AND in every query adds nothing to any kernel value,
        the outputs are independent GPs (the junk outputs are cut off anyway), and a length-scale that never meets a non-zero
        difference has no gradient.  Training rows are [z, junk] (non-zero columns 0 .. n_s + n_u), queries [states, junk,
        actions, junk] (non-zero columns 0 .. n_s and n_s + J .. n_s + J + n_u): the inner model is built over the union of
        those columns with the real outputs only -- for J >= n_u that is n_s states and 2 n_u "actions" -- and gives the
        padded model's numbers for the real outputs, the returned Jacobian's leading n_s + n_u columns included."""
        n_s, n_u, js = self.num_states, self.num_actions, self._junk_states
        keep = sorted(set(range(n_s + n_u)) | set(range(n_s + js, n_s + js + n_u)))
        try:
            inner = constructor(state_dimen=n_s, action_dimen=len(keep) - n_s)
        except ValueError:
            raise too_large from None
        if getattr(inner, 'kernel_family', None) != 'rbf':
            raise too_large     # (only the RBF exact GP is known to fold exactly)
        self._cols = torch.tensor(keep, dtype=torch.long)
        return inner

    @property
    def folded_columns(self) -> Optional[Tuple[int, ...]]:
        """The padded input columns the inner model sees (None: the inner model is the padded one, as in the reference)."""
        return None if self._cols is None else tuple(int(c) for c in self._cols)

    def _query(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor]:
        es = self._expand(states, self.num_states, self._junk_states)
        ea = self._expand(actions, self.num_actions, self._junk_actions)
        if self._cols is None:
            return es, ea
        z = torch.cat((es, ea), dim=1)[:, self._cols.to(es.device)]
        return z[:, :self.num_states], z[:, self.num_states:]

    def predict_with_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        means, variances, jacs = self._ssm.predict_with_jacobians(*self._query(states, actions))
        return (means[:, :self.num_states], variances[:, :self.num_states],
                jacs[:, :self.num_states, :(self.num_states + self.num_actions)])

    def predict_without_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor]:
        means, variances = self._ssm.predict_without_jacobians(*self._query(states, actions))
        return means[:, :self.num_states], variances[:, :self.num_states]

    def predict_raw(self, z: Tensor) -> Tuple[Tensor, Tensor]:
        ez = self._expand(z, self.num_states + self.num_actions, self._junk_states + self._junk_actions)
        if self._cols is not None:
            ez = ez[:, self._cols.to(ez.device)]
        means, variances = self._ssm.predict_raw(ez)
        return means[:, :self.num_states], variances[:, :self.num_states]

    def update_model(self, train_x: Tensor, train_y: Tensor, opt_hyp=False, replace_old=False) -> None:
        super().update_model(train_x, train_y, opt_hyp, replace_old)
        ex = self._expand(train_x, self.num_states + self.num_actions, self._junk_states + self._junk_actions)
        if self._cols is not None:
            self._ssm.update_model(ex[:, self._cols.to(ex.device)], train_y, opt_hyp, replace_old)
            return
        self._ssm.update_model(ex, self._expand(train_y, self.num_states, self._junk_states), opt_hyp, replace_old)

    @staticmethod
    def _expand(x: Tensor, real_dimen: int, junk_dimen: int) -> Tensor:
        n = x.size(0)
        assert_shape(x, (n, real_dimen))
        expanded = torch.zeros((n, real_dimen + junk_dimen), device=x.device, dtype=x.dtype)
        expanded[:, :real_dimen] = x
        return expanded

    def _update_model(self, x_train: Tensor, y_train: Tensor) -> None:
        pass

    def _train_model(self, x_train: Tensor, y_train: Tensor) -> None:
        pass

    def collect_metrics(self) -> Dict[str, Any]:
        return self._ssm.collect_metrics()

    @property
    def parametric(self) -> bool:
        return self._ssm.parametric
