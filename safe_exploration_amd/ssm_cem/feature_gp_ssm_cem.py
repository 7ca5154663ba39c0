"""Exact-GP state-space models with the reference's 'linear' and 'nn' kernels, predicted by libsxamd in WEIGHT space.

The reference's ``GpCemSSM`` builds ``ScaleKernel(LinearKernel)`` or ``ScaleKernel(NNFeatureKernel)`` for
``conf.exact_gp_kernel in ('linear', 'nn')`` (``safe_exploration/ssm_cem/gp_ssm_cem.py:45-57,140-185``) and lets
gpytorch treat them like any kernel: N x N solves.  Both are degenerate, ``k_d(x, x') = c_d phi(x) . phi(x')`` with
``c_d = outputscale_d * variance_d`` and ``phi`` the identity or a small fully connected network (ReLU between the layers,
PReLU behind the last, then the per-point min/max normalisation of :176-181), so the posterior is Bayesian linear
regression on F features (``csrc/sx_feat.hpp``): ``sx_feat_features`` -> ``sx_feat_fit`` (F x F Cholesky) on
``update_model``, ``sx_feat_predict`` / ``sx_cem_rollout_feat`` (one particle per lane) afterwards -- nothing at prediction
time depends on N_train.  The oracle (``oracle.gp.FeatureGP``) computes the same posterior in kernel space.

Hyper-parameters follow gpytorch's parameterisation (softplus of raw parameters starting at 0, noise floor 1e-4); the
network starts from ``torch.nn.Linear``'s default initialisation.  Training (``opt_hyp``): Adam, lr 0.01, on the exact
marginal log likelihood as in the reference (:103-129); its gradient w.r.t. (outputscale, variance, noise) and w.r.t. the
feature matrix is closed form from the device's fit products, only the chain rule through the network is torch autograd.
"""
import ctypes
import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import _lib
from ..utils import assert_shape, get_device
from .gp_ssm_cem import GpCemSSM, _NOISE_FLOOR


class FeatureGpCemSSM(GpCemSSM):
    kernel_family = 'feature'

    def __init__(self, conf, state_dimen: int, action_dimen: int, model=None):
        # (GpCemSSM.__new__ routes here for exact_gp_kernel in ('linear', 'nn'); its __init__ is not run)
        from .ssm_cem import CemSSM
        CemSSM.__init__(self, state_dimen, action_dimen)
        if model is not None:
            raise NotImplementedError('injecting a gpytorch model is not supported: the GP is evaluated by libsxamd')
        if state_dimen > _lib.SX_MAX_NS or action_dimen > _lib.SX_MAX_NU:
            raise ValueError(f'state/action dimension ({state_dimen}, {action_dimen}) beyond the compiled limits')
        self._kernel = getattr(conf, 'exact_gp_kernel', 'linear')
        if self._kernel not in ('linear', 'nn'):
            raise ValueError(f'Unknown kernel {self._kernel}')
        self._device = torch.device(get_device(conf))
        self._training_iterations = int(getattr(conf, 'exact_gp_training_iterations', 0))
        d_in = state_dimen + action_dimen
        self._net: Optional[nn.Sequential] = None
        if self._kernel == 'nn':
            sizes = [int(s) for s in conf.nn_kernel_layers]
            if not 1 <= len(sizes) <= _lib.SX_FEAT_MAX_LAYERS or max(sizes) > _lib.SX_FEAT_MAX_WIDTH or min(sizes) < 1:
                raise NotImplementedError(f'nn_kernel_layers={sizes}: the device kernels hold up to {_lib.SX_FEAT_MAX_LAYERS} '
                                          f'layers of up to {_lib.SX_FEAT_MAX_WIDTH} units')
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(int(getattr(conf, 'nn_kernel_seed', 0)))
            self._net = self._build_net(d_in, sizes).to(torch.float64)
            torch.random.set_rng_state(gen_state)
            self._widths = [d_in] + sizes
        else:
            self._widths = [d_in]
        self._n_feat = self._widths[-1]
        self._raw_outputscale = torch.zeros((state_dimen,), dtype=torch.float64)
        self._raw_variance = torch.zeros((state_dimen,), dtype=torch.float64)
        self._raw_noise = torch.zeros((state_dimen,), dtype=torch.float64)
        self._noise_floor = _NOISE_FLOOR
        self._last_training_losses: List[float] = []
        self._feat: Optional[_lib.SxFeatModel] = None
        self._buffers = ()
        self._stats = None

    @staticmethod
    def _build_net(in_dimen: int, layer_sizes: Sequence[int]) -> nn.Sequential:
        """Linear, (ReLU, Linear)*, PReLU -- the reference's NNFeatureKernel._build_net (gp_ssm_cem.py:157-169)."""
        layers: List[nn.Module] = []
        prev = in_dimen
        for i, size in enumerate(layer_sizes):
            if i != 0:
                layers.append(nn.ReLU())
            layers.append(nn.Linear(prev, size))
            prev = size
        layers.append(nn.PReLU())
        return nn.Sequential(*layers)

    # ---- hyper-parameters --------------------------------------------------------------------------------------
    @property
    def outputscale(self) -> Tensor:
        return F.softplus(self._raw_outputscale)

    @property
    def variance(self) -> Tensor:
        return F.softplus(self._raw_variance)

    @property
    def kernel_scale(self) -> Tensor:
        """c_d = outputscale_d * variance_d: the only combination the kernel depends on."""
        return self.outputscale * self.variance

    @property
    def lengthscale(self):
        raise AttributeError(f'the {self._kernel!r} kernel has no lengthscale')

    def set_hyperparameters(self, kernel_scale=None, noise=None, lengthscale=None, outputscale=None) -> None:
        """Explicit values: kernel_scale c [n_s] (stored as outputscale = c, variance = 1), noise [n_s]."""
        if lengthscale is not None:
            raise ValueError(f'the {self._kernel!r} kernel has no lengthscale')
        if outputscale is not None and kernel_scale is None:
            kernel_scale = outputscale
        n_s = self.num_states
        if kernel_scale is not None:
            c = torch.as_tensor(kernel_scale, dtype=torch.float64).cpu().expand(n_s).clone()
            self._raw_outputscale = self._inv_softplus(c)
            self._raw_variance = self._inv_softplus(torch.ones(n_s, dtype=torch.float64))
        if noise is not None:
            nz = torch.as_tensor(noise, dtype=torch.float64).cpu().expand(n_s).clone()
            if (nz <= 0).any():
                raise ValueError('noise must be positive')
            if (nz <= self._noise_floor).any():
                self._noise_floor = 0.0
            self._raw_noise = self._inv_softplus(nz - self._noise_floor)
        if self._x_train is not None:
            self._update_model(self._x_train, self._y_train)

    def set_network(self, layers, prelu: float = 0.25) -> None:
        """Explicit network weights: layers = [(W [out x in], b [out]), ...] (tests inject the oracle's)."""
        if self._net is None:
            raise ValueError('the linear kernel has no network')
        linears = [m for m in self._net if isinstance(m, nn.Linear)]
        if len(linears) != len(layers):
            raise ValueError(f'Wanted {len(linears)} layers, got {len(layers)}')
        with torch.no_grad():
            for lin, (W, b) in zip(linears, layers):
                lin.weight.copy_(torch.as_tensor(W, dtype=torch.float64))
                lin.bias.copy_(torch.as_tensor(b, dtype=torch.float64))
            self._net[-1].weight.fill_(float(prelu))
        if self._x_train is not None:
            self._update_model(self._x_train, self._y_train)

    def state_dict(self) -> Dict[str, Dict[str, Tensor]]:
        model = {'raw_outputscale': self._raw_outputscale.clone(), 'raw_variance': self._raw_variance.clone()}
        if self._net is not None:
            model.update({f'net_{k.replace(".", "_")}': v.detach().cpu().clone() for k, v in self._net.state_dict().items()})
        return {'gp_model': model,
                'gp_likelihood': {'raw_noise': self._raw_noise.clone(), 'noise_floor': torch.tensor(self._noise_floor)}}

    # ---- model (re)build: the warm path ------------------------------------------------------------------------
    def _net_buffer(self, dev) -> Optional[Tensor]:
        if self._net is None:
            return None
        parts = []
        for m in self._net:
            if isinstance(m, nn.Linear):
                parts += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
        return torch.cat(parts).to(dev, torch.float64).contiguous()

    def _struct(self, net_buf: Optional[Tensor]) -> _lib.SxFeatModel:
        m = _lib.SxFeatModel()
        m.n_s, m.n_u, m.n_feat = self.num_states, self.num_actions, self._n_feat
        m.n_layers = len(self._widths) - 1
        m.normalise = 1 if self._net is not None else 0
        for i, w in enumerate(self._widths):
            m.width[i] = w
        m.prelu = float(self._net[-1].weight.detach().reshape(-1)[0]) if self._net is not None else 0.0
        _lib.fill(m.noise, self.noise.numpy())
        m.net = net_buf.data_ptr() if net_buf is not None else None
        return m

    def _fit(self, x: Tensor, y: Tensor):
        """Features + weight-space fit for the current parameters: (struct, net buffer, Phi, wbar, minv, stats, status)."""
        lib = _lib.lib()
        dev = x.device
        n_s, n, Fd = self.num_states, x.size(0), self._n_feat
        net_buf = self._net_buffer(dev)
        m = self._struct(net_buf)
        phi = torch.empty((n, Fd), dtype=torch.float64, device=dev)
        _lib.check(lib.sx_feat_features(ctypes.byref(m), _lib.ptr(x), n, _lib.ptr(phi), _lib.stream_ptr(dev)), 'sx_feat_features')
        wbar = torch.empty((n_s, Fd), dtype=torch.float64, device=dev)
        minv = torch.empty((n_s, Fd, Fd), dtype=torch.float64, device=dev)
        stats = torch.empty((n_s, 3), dtype=torch.float64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        lam = (ctypes.c_double * n_s)(*[float(v) for v in (self.noise / self.kernel_scale)])
        _lib.check(lib.sx_feat_fit(ctypes.byref(m), _lib.ptr(phi), _lib.ptr(y), n, lam, _lib.ptr(wbar), _lib.ptr(minv),
                                   _lib.ptr(stats), _lib.ptr(status), _lib.stream_ptr(dev)), 'sx_feat_fit')
        m.wbar, m.minv = wbar.data_ptr(), minv.data_ptr()
        return m, net_buf, phi, wbar, minv, stats, status

    def _update_model(self, x_train: Tensor, y_train: Tensor) -> None:
        _lib.require_gpu(x_train, 'train_x')
        _lib.require_gpu(y_train, 'train_y')
        x, y = x_train.detach().contiguous(), y_train.detach().contiguous()
        m, net_buf, phi, wbar, minv, stats, status = self._fit(x, y)
        host = torch.cat((stats.reshape(-1), status.double())).cpu()
        if int(host[-1]) & _lib.SX_STATUS_NOT_PD:
            raise RuntimeError('Phi^T Phi + noise / c I is not positive definite for the current parameters')
        self._feat = m
        self._buffers = (x, net_buf, wbar, minv)
        self._stats = host[:-1].reshape(self.num_states, 3)
        self._n_train = x.size(0)
        # 1/2 log det(I + K_d / noise_d) = 1/2 log det(A_d) - F/2 log(noise_d / c_d)
        lam = self.noise / self.kernel_scale
        self._info_gain = (self._stats[:, 2] - 0.5 * self._n_feat * torch.log(lam)).numpy()

    def _mll_from_stats(self, stats: Tensor, n: int) -> Tensor:
        """Exact marginal log likelihood per output from {y^T y, |M Phi^T y|^2, sum log diag chol(A)} (Woodbury)."""
        noise, c, Fd = self.noise, self.kernel_scale, self._n_feat
        quad = (stats[:, 0] - stats[:, 1]) / noise
        logdet = (n - Fd) * torch.log(noise) + Fd * torch.log(c) + 2.0 * stats[:, 2]
        return -0.5 * quad - 0.5 * logdet - 0.5 * n * math.log(2.0 * math.pi)

    def mll(self) -> Tensor:
        """[n_s] exact marginal log likelihood of the stored data at the current parameters (host tensor)."""
        if self._feat is None:
            raise RuntimeError('the GP has no training data yet: call update_model first')
        return self._mll_from_stats(self._stats, self._n_train)

    def mll_and_grad(self, x_train: Tensor, y_train: Tensor):
        """(mll [n_s], d mll / d (kernel_scale [n_s], noise [n_s]), d sum_d mll_d / d Phi [N x F]) at the current parameters.
        Closed forms in the fit products (alpha = K^-1 y, A = Phi^T Phi + lambda I):
            Phi^T alpha = wbar / c      K^-1 Phi = Phi A^-1 / c      tr K^-1 = (N - F + lambda tr A^-1) / noise
            d mll / d c = 1/2 (|wbar|^2 / c^2 - (F - lambda tr A^-1) / c)        d mll / d noise = 1/2 (|alpha|^2 - tr K^-1)
            d mll / d Phi = alpha wbar^T - Phi A^-1
        """
        x, y = x_train.detach().contiguous(), y_train.detach().contiguous()
        m, net_buf, phi, wbar, minv, stats, status = self._fit(x, y)
        n, Fd = x.size(0), self._n_feat
        dev = x.device
        noise, c = self.noise.to(dev), self.kernel_scale.to(dev)
        lam = noise / c
        ainv = minv.transpose(1, 2) @ minv                                        # [n_s x F x F]  A^-1 = M^T M
        tr_ainv = ainv.diagonal(dim1=1, dim2=2).sum(1)
        resid = y - phi @ wbar.t()                                                # [N x n_s]
        alpha = resid / noise                                                     # K^-1 y
        d_c = 0.5 * ((wbar * wbar).sum(1) / c ** 2 - (Fd - lam * tr_ainv) / c)
        d_noise = 0.5 * ((alpha * alpha).sum(0) - (n - Fd + lam * tr_ainv) / noise)
        d_phi = alpha @ wbar - torch.einsum('nf,dfg->ng', phi, ainv)              # summed over the outputs
        host = torch.cat((stats.reshape(-1), status.double())).cpu()
        if int(host[-1]) & _lib.SX_STATUS_NOT_PD:
            raise RuntimeError('Phi^T Phi + noise / c I is not positive definite for the current parameters')
        return self._mll_from_stats(host[:-1].reshape(self.num_states, 3), n), d_c.cpu(), d_noise.cpu(), d_phi

    def _features_torch(self, x: Tensor) -> Tensor:
        """phi(x) with torch ops (autograd): only the training loop uses it, for the chain rule through the network."""
        f = self._net(x)
        mn = f.min(dim=1, keepdim=True)[0]
        return 2.0 * ((f - mn) / f.max(dim=1, keepdim=True)[0]) - 1.0

    def _train_model(self, x_train: Tensor, y_train: Tensor) -> None:
        if self._training_iterations <= 0:
            return
        n = x_train.size(0)
        raw = [self._raw_outputscale.clone().requires_grad_(True), self._raw_variance.clone().requires_grad_(True),
               self._raw_noise.clone().requires_grad_(True)]
        params = [{'params': raw}]
        if self._net is not None:
            self._net.to(x_train.device)
            params.append({'params': list(self._net.parameters())})
        opt = torch.optim.Adam(params, lr=0.01)
        losses = []
        for _ in range(self._training_iterations):
            opt.zero_grad()
            self._raw_outputscale, self._raw_variance, self._raw_noise = (t.detach() for t in raw)
            mll, d_c, d_noise, d_phi = self.mll_and_grad(x_train, y_train)
            losses.append(float(-(mll / n).sum()))
            # loss = -sum_d mll_d / N (gpytorch's ExactMarginalLogLikelihood divides by N); c = softplus(a) softplus(b)
            s, v = self.outputscale, self.variance
            raw[0].grad = -(d_c * v / n) * torch.sigmoid(raw[0].detach())
            raw[1].grad = -(d_c * s / n) * torch.sigmoid(raw[1].detach())
            raw[2].grad = -(d_noise / n) * torch.sigmoid(raw[2].detach())
            if self._net is not None:
                (-(d_phi.detach() / n) * self._features_torch(x_train.detach())).sum().backward()
            opt.step()
        self._raw_outputscale, self._raw_variance, self._raw_noise = (t.detach().clone() for t in raw)
        self._last_training_losses = losses
        self._update_model(x_train, y_train)

    def information_gain(self):
        import numpy as np
        return np.zeros(self.num_states) if self._feat is None else self._info_gain.copy()

    # ---- prediction: the hot path ------------------------------------------------------------------------------
    @property
    def device_model(self):
        raise RuntimeError('the degenerate-kernel GP has no sx_gp_model: use feat_model (sx_cem_rollout_feat)')

    @property
    def feat_model(self) -> _lib.SxFeatModel:
        if self._feat is None:
            raise RuntimeError('the GP has no training data yet: call update_model first')
        return self._feat

    def workspace(self, nbytes: int):
        return None

    def _predict_z(self, z: Tensor, jacobians: bool) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        n, d_in = z.size(0), self.num_states + self.num_actions
        assert_shape(z, (n, d_in))
        _lib.require_gpu(z, 'states/actions')
        z = z.detach().contiguous()
        mean = torch.empty((n, self.num_states), dtype=torch.float64, device=z.device)
        var = torch.empty_like(mean)
        jac = torch.empty((n, self.num_states, d_in), dtype=torch.float64, device=z.device) if jacobians else None
        if n == 0:
            return mean, var, jac
        if self._feat is None:
            raise RuntimeError('the GP has no training data yet: call update_model first')
        _lib.check(_lib.lib().sx_feat_predict(ctypes.byref(self._feat), _lib.ptr(z), n, _lib.ptr(mean), _lib.ptr(var),
                                              _lib.ptr(jac), _lib.stream_ptr(z.device)), 'sx_feat_predict')
        return mean, var, jac

    def predict_variance_jacobian(self, states: Tensor, actions: Tensor) -> Tensor:
        raise NotImplementedError('the variance Jacobian is provided for the RBF kernel only')

    def predict_mean_hessian(self, states: Tensor, actions: Tensor) -> Tensor:
        raise NotImplementedError('the mean Hessian is provided for the RBF kernel only')

    def collect_metrics(self) -> Dict[str, Any]:
        return {'losses': self._last_training_losses}
