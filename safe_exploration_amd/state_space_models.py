"""numpy-facing state-space-model interface (`StateSpaceModel.predict`) over the HIP GP.

The reference's numpy code (``gp_reachability.onestep_reachability``, the casadi solvers) talks to models through
``StateSpaceModel`` (``safe_exploration/state_space_models.py:16-213``); its torch GP is wrapped by ``GPyTorchSSM``
(``ssm_pytorch/gaussian_process.py:143-343``).  ``HipGpStateSpaceModel`` is that adapter for ``GpCemSSM``: numpy in,
numpy out, one ``sx_gp_predict`` launch per call.  The Jacobian of the VARIANCE and the reverse-mode hooks are only
consumed by the casadi callback machinery (``CasadiSSMEvaluator``, out of scope) and are not provided.
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

from .ssm_cem.gp_ssm_cem import GpCemSSM


class StateSpaceModel(ABC):
    """x_{t+1} = f(x_t, u_t) with uncertainty; states [N x n], actions [N x m] as numpy arrays."""

    def __init__(self, num_states, num_actions, has_jacobian=True, has_reverse=False):
        self.num_states = num_states
        self.num_actions = num_actions
        self.has_jacobian = has_jacobian
        self.has_reverse = has_reverse

    def __call__(self, states, actions):
        """(mean, variance, jacobian of the mean) -- what the numpy reachability code asks for."""
        return self.predict(states, actions, True, False)

    @abstractmethod
    def predict(self, states, actions, jacobians=False, full_cov=False):
        """mean [N x n], variance [N x n] (+ jacobian of the mean [N x n x (n + m)] when `jacobians`)."""

    @abstractmethod
    def update_model(self, train_x, train_y, opt_hyp=False, replace_old=False):
        """train_x [N x (n + m)], train_y [N x n]."""

    def linearize_predict(self, states, actions, jacobians=False, full_cov=False):
        raise NotImplementedError

    def get_reverse(self, seed):
        raise NotImplementedError

    def get_linearize_reverse(self, seed):
        raise NotImplementedError


class HipGpStateSpaceModel(StateSpaceModel):
    def __init__(self, ssm: GpCemSSM, device='cuda:0'):
        super().__init__(ssm.num_states, ssm.num_actions, has_jacobian=True, has_reverse=False)
        self._ssm = ssm
        self._device = torch.device(device)

    def _t(self, x):
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=self._device)

    def predict(self, states, actions, jacobians=False, full_cov=False):
        if full_cov:
            raise NotImplementedError('full covariance between query points is not computed on this path')
        states, actions = np.atleast_2d(states), np.atleast_2d(actions)
        if jacobians:
            # (mean, var, jac_mean, jac_var) like GPyTorchSSM._predict (ssm_pytorch/gaussian_process.py:222-231)
            st, ac = self._t(states), self._t(actions)
            mean, var, jac = self._ssm.predict_with_jacobians(st, ac)
            jac_var = self._ssm.predict_variance_jacobian(st, ac)
            return mean.cpu().numpy(), var.cpu().numpy(), jac.cpu().numpy(), jac_var.cpu().numpy()
        mean, var = self._ssm.predict_without_jacobians(self._t(states), self._t(actions))
        return mean.cpu().numpy(), var.cpu().numpy()

    def update_model(self, train_x, train_y, opt_hyp=False, replace_old=False):
        self._ssm.update_model(self._t(train_x), self._t(train_y), opt_hyp, replace_old)
