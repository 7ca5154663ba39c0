"""Exact-GP state-space model for the CEM safe-MPC solver, predicted by the HIP kernels of libsxamd.

Same surface as the reference's ``GpCemSSM`` (``safe_exploration/ssm_cem/gp_ssm_cem.py:17-137``): ``n_s`` independent
exact GPs on shared inputs, zero mean, scaled ARD-RBF kernel, Gaussian likelihood whose noise is INCLUDED in the
predictive variance.  The reference delegates the arithmetic to gpytorch 0.3.2; here

* ``_update_model`` factorises ``K_d + noise_d I = L_d L_d^T`` and lays ``W_d = L_d^-1`` and ``alpha_d`` out for the
  f64 matrix cores (``sx_gp_pack``) -- this is the warm path, run once per ``update_model``;
* every ``predict_*`` is one ``sx_gp_predict`` launch (mean, variance and the analytic mean-Jacobian in one pass,
  where the reference needs two forward passes and a backward pass, ``gp_ssm_cem.py:59-73``).

Hyper-parameters follow gpytorch's parameterisation (softplus of a raw parameter that starts at 0; the noise has a
1e-4 floor).  gpytorch is not available to check these defaults against: "parity unpinned" (DESIGN.md).
"""
import ctypes
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .. import _lib
from ..utils import assert_shape, get_device
from .ssm_cem import CemSSM

_NOISE_FLOOR = 1e-4


# csrc/sx_fit.hpp kFitMaxN: the factorisation's last stage keeps one column of the system in LDS
MAX_TRAINING_POINTS = 4096


class GpCemSSM(CemSSM):
    kernel_family = 'rbf'   # 'feature' for the degenerate kernels ('linear', 'nn'): feature_gp_ssm_cem.FeatureGpCemSSM

    def __new__(cls, conf=None, *args, **kwargs):
        # one class name for every kernel, as in the reference (gp_ssm_cem.py:45-57): the 'linear' and 'nn' kernels are
        # served by the weight-space subclass
        if cls is GpCemSSM and getattr(conf, 'exact_gp_kernel', 'rbf') in ('linear', 'nn'):
            from .feature_gp_ssm_cem import FeatureGpCemSSM
            return super().__new__(FeatureGpCemSSM)
        return super().__new__(cls)

    def __init__(self, conf, state_dimen: int, action_dimen: int, model=None):
        super().__init__(state_dimen, action_dimen)
        if model is not None:
            raise NotImplementedError('injecting a gpytorch model is not supported: the GP is evaluated by libsxamd')
        kernel = getattr(conf, 'exact_gp_kernel', 'rbf')
        if kernel != 'rbf':
            raise ValueError(f'Unknown kernel {kernel}')
        if state_dimen > _lib.SX_MAX_NS or action_dimen > _lib.SX_MAX_NU:
            raise ValueError(f'state/action dimension ({state_dimen}, {action_dimen}) beyond the compiled limits')
        self._device = torch.device(get_device(conf))
        self._training_iterations = int(getattr(conf, 'exact_gp_training_iterations', 0))
        d_in = state_dimen + action_dimen
        self._raw_lengthscale = torch.zeros((state_dimen, d_in), dtype=torch.float64)
        self._raw_outputscale = torch.zeros((state_dimen,), dtype=torch.float64)
        self._raw_noise = torch.zeros((state_dimen,), dtype=torch.float64)
        self._noise_floor = _NOISE_FLOOR
        self._last_training_losses = []
        self._model: Optional[_lib.SxGpModel] = None
        self._buffers = ()  # keeps the device operands alive while the struct points at them
        self._workspace: Optional[Tensor] = None
        self._fit_ws = None
        self._linv_current = False   # does _fit_ws[1] hold W of the CURRENT model?

    # ---- hyper-parameters --------------------------------------------------------------------------------------
    @property
    def lengthscale(self) -> Tensor:
        return F.softplus(self._raw_lengthscale)

    @property
    def outputscale(self) -> Tensor:
        return F.softplus(self._raw_outputscale)

    @property
    def noise(self) -> Tensor:
        return F.softplus(self._raw_noise) + self._noise_floor

    @staticmethod
    def _inv_softplus(x: Tensor) -> Tensor:
        return x + torch.log(-torch.expm1(-x))

    def set_hyperparameters(self, lengthscale=None, outputscale=None, noise=None) -> None:
        """Explicit hyper-parameters: lengthscale [n_s x D] (or broadcastable), outputscale [n_s], noise [n_s]."""
        n_s, d_in = self.num_states, self.num_states + self.num_actions
        if lengthscale is not None:
            ls = torch.as_tensor(lengthscale, dtype=torch.float64).cpu().expand(n_s, d_in).clone()
            self._raw_lengthscale = self._inv_softplus(ls)
        if outputscale is not None:
            s = torch.as_tensor(outputscale, dtype=torch.float64).cpu().expand(n_s).clone()
            self._raw_outputscale = self._inv_softplus(s)
        if noise is not None:
            nz = torch.as_tensor(noise, dtype=torch.float64).cpu().expand(n_s).clone()
            if (nz <= 0).any():
                raise ValueError('noise must be positive')
            if (nz <= self._noise_floor).any():
                self._noise_floor = 0.0   # explicit values below gpytorch's default floor: drop the floor
            self._raw_noise = self._inv_softplus(nz - self._noise_floor)
        if self._x_train is not None:
            self._update_model(self._x_train, self._y_train)

    def state_dict(self) -> Dict[str, Dict[str, Tensor]]:
        """The hyper-parameter state, split the way the reference's failure dump stores it (model / likelihood:
        gp_reachability_pytorch.py:256-266); `load_state_dict` takes it back."""
        return {'gp_model': {'raw_lengthscale': self._raw_lengthscale.clone(), 'raw_outputscale': self._raw_outputscale.clone()},
                'gp_likelihood': {'raw_noise': self._raw_noise.clone(), 'noise_floor': torch.tensor(self._noise_floor)}}

    def load_state_dict(self, state: Dict[str, Dict[str, Tensor]]) -> None:
        self._raw_lengthscale = state['gp_model']['raw_lengthscale'].clone()
        self._raw_outputscale = state['gp_model']['raw_outputscale'].clone()
        self._raw_noise = state['gp_likelihood']['raw_noise'].clone()
        self._noise_floor = float(state['gp_likelihood']['noise_floor'])
        if self._x_train is not None:
            self._update_model(self._x_train, self._y_train)

    # ---- model (re)build: the warm path ------------------------------------------------------------------------
    def _fit(self, x: Tensor, y: Tensor):
        """sx_gp_fit for the current hyper-parameters.  Returns (model struct, linv, alpha, logdet); raises if the
        kernel matrix is not positive definite."""
        lib = _lib.lib()
        dev = x.device
        n_s, n_u, n = self.num_states, self.num_actions, x.size(0)
        m = _lib.SxGpModel()
        m.n_s, m.n_u, m.n_train = n_s, n_u, n
        _lib.fill(m.inv_ls2, (1.0 / self.lengthscale ** 2).numpy())
        _lib.fill(m.outputscale, self.outputscale.numpy())
        _lib.fill(m.noise, self.noise.numpy())
        m.x_train = x.data_ptr()
        if self._fit_ws is None or self._fit_ws[0].size(1) != n or self._fit_ws[0].device != dev:
            self._fit_ws = (torch.empty((n_s, n, n), dtype=torch.float64, device=dev),
                            torch.empty((n_s, n, n), dtype=torch.float64, device=dev))
        work, linv = self._fit_ws
        self._linv_current = False   # (set again by _update_model, whose fit is the model's)
        alpha = torch.empty((n_s, n), dtype=torch.float64, device=dev)
        logdet = torch.empty((n_s,), dtype=torch.float64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.sx_gp_fit(ctypes.byref(m), _lib.ptr(y), _lib.ptr(work), _lib.ptr(linv), _lib.ptr(alpha),
                                 _lib.ptr(logdet), _lib.ptr(status), _lib.stream_ptr(dev)), 'sx_gp_fit')
        return m, linv, alpha, logdet, status

    def _update_model(self, x_train: Tensor, y_train: Tensor) -> None:
        n_s, n_u, n = self.num_states, self.num_actions, x_train.size(0)
        if n > MAX_TRAINING_POINTS:
            raise ValueError(f'{n} training points: the exact-GP kernels hold up to {MAX_TRAINING_POINTS} (sx_gp_fit); keep the '
                             f'most recent / most informative ones (update_model(..., replace_old=True))')
        _lib.require_gpu(x_train, 'train_x')
        _lib.require_gpu(y_train, 'train_y')
        lib = _lib.lib()
        dev = x_train.device
        x = x_train.detach().contiguous()
        y = y_train.detach().contiguous()
        a_n, t_n = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.sx_gp_pack_sizes(n_s, n_u, n, ctypes.byref(a_n), ctypes.byref(t_n)), 'sx_gp_pack_sizes')
        a_pack = torch.empty(a_n.value, dtype=torch.float64, device=dev)
        stage_tab = torch.empty(t_n.value, dtype=torch.int32, device=dev)
        # factorise on the device (sx_gp_fit), then lay the operands out for the matrix cores (sx_gp_pack)
        m, linv, alpha, logdet, status = self._fit(x, y)
        m.a_pack, m.stage_tab = a_pack.data_ptr(), stage_tab.data_ptr()
        _lib.check(lib.sx_gp_pack(ctypes.byref(m), _lib.ptr(linv), _lib.ptr(alpha), _lib.stream_ptr(dev)), 'sx_gp_pack')
        if int(status.item()) & _lib.SX_STATUS_NOT_PD:
            raise RuntimeError('the kernel matrix K + noise I is not positive definite for the current hyper-parameters')
        self._model = m
        self._buffers = (x, a_pack, stage_tab)
        self._alpha = alpha
        self._linv_current = True
        # 1/2 log det(I + K_d / noise_d) = sum log diag L_d - N/2 log noise_d
        self._info_gain = (logdet.cpu() - 0.5 * n * torch.log(self.noise)).numpy()

    def mll_and_grad(self, x_train: Tensor, y_train: Tensor):
        """Exact marginal log likelihood per output [n_s] and its gradient [n_s x (D + 2)] w.r.t.
        (lengthscale, outputscale, noise), at the current hyper-parameters (host tensors)."""
        x, y = x_train.detach().contiguous(), y_train.detach().contiguous()
        m, linv, alpha, logdet, status = self._fit(x, y)
        d_in = self.num_states + self.num_actions
        mll = torch.empty((self.num_states,), dtype=torch.float64, device=x.device)
        grad = torch.empty((self.num_states, d_in + 2), dtype=torch.float64, device=x.device)
        _lib.check(_lib.lib().sx_gp_mll_grad(ctypes.byref(m), _lib.ptr(y), _lib.ptr(linv), _lib.ptr(alpha), _lib.ptr(logdet),
                                             _lib.ptr(self._fit_ws[0]), _lib.ptr(mll), _lib.ptr(grad),
                                             _lib.stream_ptr(x.device)), 'sx_gp_mll_grad')
        host = torch.cat((mll, grad.reshape(-1), status.double())).cpu()
        if int(host[-1]) & _lib.SX_STATUS_NOT_PD:
            raise RuntimeError('the kernel matrix K + noise I is not positive definite for the current hyper-parameters')
        return host[:self.num_states], host[self.num_states:-1].reshape(self.num_states, d_in + 2)

    def information_gain(self):
        """[n_s] information gain of the training inputs, per output (zeros without data)."""
        import numpy as np
        return np.zeros(self.num_states) if self._model is None else self._info_gain.copy()

    def workspace(self, nbytes: int) -> Optional[Tensor]:
        """Scratch for the large-training-set rollout path (grown on demand, reused across calls)."""
        if nbytes <= 0:
            return None
        if self._workspace is None or self._workspace.numel() * 8 < nbytes:
            self._workspace = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=self._buffers[0].device)
        return self._workspace

    @property
    def device_model(self) -> _lib.SxGpModel:
        """The sx_gp_model the fused solver hands to sx_cem_rollout."""
        if self._model is None:
            raise RuntimeError('the GP has no training data yet: call update_model first')
        return self._model

    def _train_model(self, x_train: Tensor, y_train: Tensor) -> None:
        """Adam (lr 0.01) on the exact marginal log likelihood, the reference's recipe (gp_ssm_cem.py:103-129): the
        loss is -sum_d mll_d / N (gpytorch's ExactMarginalLogLikelihood divides by the number of data points).
        Value and gradient come from the device (sx_gp_fit + sx_gp_mll_grad); the few-parameter Adam step and the
        softplus chain rule stay on the host."""
        if self._training_iterations <= 0:
            return
        n = x_train.size(0)
        d_in = self.num_states + self.num_actions
        raw = [self._raw_lengthscale.clone().requires_grad_(True), self._raw_outputscale.clone().requires_grad_(True),
               self._raw_noise.clone().requires_grad_(True)]
        opt = torch.optim.Adam(raw, lr=0.01)
        losses = []
        for _ in range(self._training_iterations):
            self._raw_lengthscale, self._raw_outputscale, self._raw_noise = (t.detach() for t in raw)
            mll, grad = self.mll_and_grad(x_train, y_train)
            losses.append(float(-(mll / n).sum()))
            # d loss / d raw = -(1/N) d mll / d theta * sigmoid(raw)   (theta = softplus(raw) [+ floor])
            raw[0].grad = -(grad[:, :d_in] / n) * torch.sigmoid(raw[0].detach())
            raw[1].grad = -(grad[:, d_in] / n) * torch.sigmoid(raw[1].detach())
            raw[2].grad = -(grad[:, d_in + 1] / n) * torch.sigmoid(raw[2].detach())
            opt.step()
        self._raw_lengthscale, self._raw_outputscale, self._raw_noise = (t.detach().clone() for t in raw)
        self._last_training_losses = losses
        self._update_model(x_train, y_train)

    # ---- prediction: the hot path ------------------------------------------------------------------------------
    def _predict_z(self, z: Tensor, jacobians: bool) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        n = z.size(0)
        d_in = self.num_states + self.num_actions
        assert_shape(z, (n, d_in))
        _lib.require_gpu(z, 'states/actions')
        z = z.detach().contiguous()
        mean = torch.empty((n, self.num_states), dtype=torch.float64, device=z.device)
        var = torch.empty_like(mean)
        jac = torch.empty((n, self.num_states, d_in), dtype=torch.float64, device=z.device) if jacobians else None
        if n == 0:
            return mean, var, jac
        if self._model is None:
            # no data: the prior (zero mean, s + noise variance, flat mean)
            mean.zero_()
            var.copy_((self.outputscale + self.noise).to(z.device).expand(n, -1))
            if jac is not None:
                jac.zero_()
            return mean, var, jac
        lib = _lib.lib()
        ws_bytes = int(lib.sx_gp_predict_workspace_bytes(ctypes.byref(self._model), n))
        _lib.check(lib.sx_gp_predict(ctypes.byref(self._model), _lib.ptr(z), n, _lib.ptr(mean), _lib.ptr(var),
                                     _lib.ptr(jac), _lib.ptr(self.workspace(ws_bytes)), ws_bytes,
                                     _lib.stream_ptr(z.device)), 'sx_gp_predict')
        return mean, var, jac

    def predict_with_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        return self._predict_z(self._join_states_actions(states, actions), True)

    def predict_without_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor]:
        mean, var, _ = self._predict_z(self._join_states_actions(states, actions), False)
        return mean, var

    def predict_variance_jacobian(self, states: Tensor, actions: Tensor) -> Tensor:
        """d var / d (state, action)  [N x n_s x (n_s + n_u)].  Not used by the CEM solver; the numpy adapter
        (state_space_models.HipGpStateSpaceModel) returns it where the reference's GPyTorchSSM returns
        ``compute_jacobian(pred_var, inp)`` (ssm_pytorch/gaussian_process.py:222-231)."""
        z = self._join_states_actions(states, actions)
        n, d_in = z.size(0), self.num_states + self.num_actions
        _lib.require_gpu(z, 'states/actions')
        z = z.detach().contiguous()
        out = torch.zeros((n, self.num_states, d_in), dtype=torch.float64, device=z.device)
        if n == 0 or self._model is None:   # the prior variance does not depend on z
            return out
        if not self._linv_current:
            # the fit workspace was reused since (hyper-parameter training): factorise once more
            self._fit(self._x_train.detach().contiguous(), self._y_train.detach().contiguous())
            self._linv_current = True
        _lib.check(_lib.lib().sx_gp_predict_var_jac(ctypes.byref(self._model), _lib.ptr(self._fit_ws[1]), _lib.ptr(z), n,
                                                   _lib.ptr(out), _lib.stream_ptr(z.device)), 'sx_gp_predict_var_jac')
        return out

    def predict_mean_hessian(self, states: Tensor, actions: Tensor) -> Tensor:
        """d^2 mean / d(state, action)^2  [N x n_s x (n_s + n_u) x (n_s + n_u)] (closed form, sx_gp_predict_mean_hessian).
        Not used by the CEM solver; the numpy adapter's linearize_predict returns it where the reference's GPyTorchSSM
        calls the `hessian` package (ssm_pytorch/gaussian_process.py:160-187)."""
        z = self._join_states_actions(states, actions)
        n, d_in = z.size(0), self.num_states + self.num_actions
        _lib.require_gpu(z, 'states/actions')
        z = z.detach().contiguous()
        out = torch.zeros((n, self.num_states, d_in, d_in), dtype=torch.float64, device=z.device)
        if n == 0 or self._model is None:   # the prior mean is flat
            return out
        _lib.check(_lib.lib().sx_gp_predict_mean_hessian(ctypes.byref(self._model), _lib.ptr(self._alpha), _lib.ptr(z), n,
                                                        _lib.ptr(out), _lib.stream_ptr(z.device)), 'sx_gp_predict_mean_hessian')
        return out

    def predict_raw(self, z: Tensor) -> Tuple[Tensor, Tensor]:
        mean, var, _ = self._predict_z(z, False)
        return mean.t(), var.t()

    def collect_metrics(self) -> Dict[str, Any]:
        return {'losses': self._last_training_losses}

    @property
    def parametric(self) -> bool:
        return False
