"""numpy-facing state-space-model interface (`StateSpaceModel.predict`) over the HIP GP.

The reference's numpy code (``gp_reachability.onestep_reachability``, the casadi solvers) talks to models through
``StateSpaceModel`` (``safe_exploration/state_space_models.py:16-213``); its torch GP is wrapped by ``GPyTorchSSM``
(``ssm_pytorch/gaussian_process.py:143-343``).  ``HipGpStateSpaceModel`` is that adapter for ``GpCemSSM``: numpy in,
numpy out, one ``sx_gp_predict`` launch per call.  ``linearize_predict`` (mean Hessian) and the reverse-mode hooks
``get_reverse`` / ``get_linearize_reverse`` -- what ``CasadiSSMEvaluator`` (default ``linearize_mu=True``,
``state_space_models.py:279-304``) asks of a model -- are served from closed forms (``sx_gp_predict_var_jac``,
``sx_gp_predict_mean_hessian``) where the reference back-propagates through gpytorch.
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
        super().__init__(ssm.num_states, ssm.num_actions, has_jacobian=True, has_reverse=True)
        self._ssm = ssm
        self._device = torch.device(device)
        self._last_inputs = None          # (states, actions) of the last predict: what get_reverse differentiates at
        self._last_linearize_inputs = None

    def _t(self, x):
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=self._device)

    def _all(self, states, actions):
        st, ac = self._t(np.atleast_2d(states)), self._t(np.atleast_2d(actions))
        mean, var, jac = self._ssm.predict_with_jacobians(st, ac)
        jac_var = self._ssm.predict_variance_jacobian(st, ac)
        return st, ac, mean.cpu().numpy(), var.cpu().numpy(), jac.cpu().numpy(), jac_var.cpu().numpy()

    def predict(self, states, actions, jacobians=False, full_cov=False):
        if full_cov:
            raise NotImplementedError('full covariance between query points is not computed on this path')
        states, actions = np.atleast_2d(states), np.atleast_2d(actions)
        self._last_inputs = (states.copy(), actions.copy())
        if jacobians:
            # (mean, var, jac_mean, jac_var) like GPyTorchSSM._predict (ssm_pytorch/gaussian_process.py:222-231)
            return self._all(states, actions)[2:]
        mean, var = self._ssm.predict_without_jacobians(self._t(states), self._t(actions))
        return mean.cpu().numpy(), var.cpu().numpy()

    def linearize_predict(self, states, actions, jacobians=False, full_cov=False):
        """(mean, var, jac_mean) -- the quantities of the first-order expansion of the mean -- and with `jacobians`
        also (jac_var, hess_mean [n x (n + m) x (n + m)]): reference ssm_pytorch/gaussian_process.py:270-316.  Like the
        reference, the second-order outputs are for a single input row."""
        if full_cov:
            raise NotImplementedError('full covariance between query points is not computed on this path')
        states, actions = np.atleast_2d(states), np.atleast_2d(actions)
        if jacobians and states.shape[0] > 1:
            raise NotImplementedError("'linearize_predict' only allows single inputs, i.e. (1 x n) arrays, when computing "
                                      "jacobians.")
        st, ac, mean, var, jac_mean, jac_var = self._all(states, actions)
        self._last_linearize_inputs = (states.copy(), actions.copy())
        if not jacobians:
            return mean, var, jac_mean
        hess = self._ssm.predict_mean_hessian(st, ac)[0].cpu().numpy()
        return mean, var, jac_mean, jac_var, hess

    def get_reverse(self, seed):
        """Reverse-mode derivative of the last `predict`'s outputs: seed [2 n x N] over (mean; var) -> (grad_state [n],
        grad_action [m]) at the FIRST input row (reference :318-327, which reads ``inp.grad[0]``)."""
        if self._last_inputs is None:
            raise RuntimeError('get_reverse needs a preceding predict')
        n = self.num_states
        _, _, _, _, jac_mean, jac_var = self._all(self._last_inputs[0][:1], self._last_inputs[1][:1])
        seed = np.asarray(seed, dtype=np.float64).reshape(2 * n, -1)[:, 0]
        grad = jac_mean[0].T @ seed[:n] + jac_var[0].T @ seed[n:]
        return grad[:n], grad[n:]

    def get_linearize_reverse(self, seed):
        """The same for `linearize_predict`'s (mean; var; vec(jac_mean)): seed [(2 n + n (n + m)) x 1] ->
        (grad_state [n x 1], grad_action [m x 1]) (reference :329-336); the Jacobian block needs the mean Hessian."""
        if self._last_linearize_inputs is None:
            raise RuntimeError('get_linearize_reverse needs a preceding linearize_predict')
        n, d_in = self.num_states, self.num_states + self.num_actions
        st, ac, _, _, jac_mean, jac_var = self._all(self._last_linearize_inputs[0][:1], self._last_linearize_inputs[1][:1])
        hess = self._ssm.predict_mean_hessian(st, ac)[0].cpu().numpy()          # [n x D x D]
        seed = np.asarray(seed, dtype=np.float64).reshape(-1)
        if seed.size != 2 * n + n * d_in:
            raise ValueError(f'Wanted a seed of {2 * n + n * d_in} entries, got {seed.size}')
        s_jac = seed[2 * n:].reshape(n, d_in)
        grad = jac_mean[0].T @ seed[:n] + jac_var[0].T @ seed[n:2 * n] + np.einsum('dj,djl->l', s_jac, hess)
        return grad[:n, None], grad[n:, None]

    def update_model(self, train_x, train_y, opt_hyp=False, replace_old=False):
        self._ssm.update_model(self._t(train_x), self._t(train_y), opt_hyp, replace_old)
