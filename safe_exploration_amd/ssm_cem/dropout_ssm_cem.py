"""MC-dropout state-space models for the CEM safe-MPC solver, predicted by libsxamd as an ensemble of thinned networks.

Mirrors the surface of the reference's ``McDropoutSSM`` (``safe_exploration/ssm_cem/dropout_ssm_cem.py``): same config keys
(``mc_dropout_training_iterations``, ``mc_dropout_num_samples``, ``mc_dropout_hidden_features``, ``mc_dropout_type`` 'fixed' |
'concrete', ``mc_dropout_fixed_probability``, ``mc_dropout_concrete_initial_probability``, ``mc_dropout_on_input``,
``mc_dropout_predict_std``, ``mc_dropout_reinitialize``), parametric (``update_model`` always trains), metrics = the layers'
dropout probabilities.

The reference builds its network with the third-party ``bnn`` package (absent here: layer layout, regulariser and the
eval-time masks cannot be read; **parity unpinned**) and samples predictions through torch's RNG.  Here:

* the network is ``Linear -> ReLU -> dropout`` per hidden layer (+ input dropout when ``mc_dropout_on_input``) and a linear
  output layer; training (the warm path) is torch autograd on the device, Adam as in the reference (:120-150), fresh masks per
  step, loss = MSE (fixed) or Gaussian negative log likelihood (concrete) + 1e-2 x regulariser (:152-172; the regulariser is
  this module's: squared weights / (1 - p) and, for concrete dropout, the entropy term of Gal et al.);
* after training, ``mc_dropout_num_samples`` mask sets are drawn ONCE (seeded) and frozen: prediction is the deterministic
  ensemble of ``csrc/sx_mlp.hpp`` -- mean and unbiased variance over the members (:100-112), mean Jacobian by reverse sweeps
  where the reference differentiates a second stochastic forward pass with autograd (:79-90);
* with ``mc_dropout_predict_std`` the variance gains the members' mean ``exp(2 log_std)``: the expectation, over the
  reference's fresh ``randn`` aleatoric noise, of the sample variance it computes (:106-109).
"""
import ctypes
import math
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib
from ..utils import assert_shape, get_device
from .ssm_cem import CemSSM

_TEMPERATURE = 0.1   # concrete relaxation (gal_concrete_dropout.py:63)
_EPS = 1e-7


class _DropoutNet(nn.Module):
    """Hidden layers with (optionally learnable) dropout rates; masks multiply a layer's INPUT units."""

    def __init__(self, in_features: int, out_features: int, hidden: List[int], rate: float, concrete: bool, on_input: bool):
        super().__init__()
        sizes = [in_features] + list(hidden)
        self.linears = nn.ModuleList(nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(hidden)))
        self.out = nn.Linear(sizes[-1], out_features)
        self.concrete = concrete
        self.on_input = on_input
        logit = math.log(rate) - math.log(1.0 - rate)
        # one rate per mask (input + every hidden layer); learnable for concrete dropout
        self.p_logit = nn.Parameter(torch.full((len(sizes),), logit), requires_grad=concrete)
        self.sizes = sizes

    def rates(self) -> Tensor:
        return torch.sigmoid(self.p_logit)

    def draw_masks(self, shape_prefix: Tuple[int, ...], generator: Optional[torch.Generator] = None) -> List[Tensor]:
        """Multipliers for the input and every hidden layer: Bernoulli(1 - p) / (1 - p), or the concrete relaxation."""
        p = self.rates()
        dev = self.p_logit.device
        masks = []
        for i, width in enumerate(self.sizes):
            if i == 0 and not self.on_input:
                masks.append(torch.ones(shape_prefix + (width,), dtype=torch.float64, device=dev))
                continue
            u = torch.rand(shape_prefix + (width,), dtype=torch.float64, device=dev, generator=generator)
            if self.concrete:
                drop = torch.sigmoid((torch.log(p[i] + _EPS) - torch.log(1 - p[i] + _EPS) + torch.log(u + _EPS)
                                      - torch.log(1 - u + _EPS)) / _TEMPERATURE)
                masks.append((1 - drop) / (1 - p[i]))
            else:
                masks.append((u >= p[i]).to(torch.float64) / (1 - p[i]))
        return masks

    def forward(self, x: Tensor, masks: List[Tensor]) -> Tensor:
        a = x * masks[0]
        for lin, m in zip(self.linears, masks[1:]):
            a = torch.relu(lin(a)) * m
        return self.out(a)

    def regularization(self, n_data: int) -> Tensor:
        p = self.rates()
        reg = torch.zeros((), dtype=torch.float64, device=self.p_logit.device)
        layers = list(self.linears) + [self.out]
        for i, lin in enumerate(layers):
            pi = p[i] if (i > 0 or self.on_input) else torch.zeros_like(p[i])
            reg = reg + (lin.weight.pow(2).sum() + lin.bias.pow(2).sum()) / (1 - pi) / n_data
            if self.concrete and (i > 0 or self.on_input):
                reg = reg + (pi * torch.log(pi) + (1 - pi) * torch.log(1 - pi)) * self.sizes[i] * 2.0 / n_data
        return reg


class McDropoutSSM(CemSSM):
    kernel_family = 'mlp'

    def __init__(self, conf, state_dimen: int, action_dimen: int):
        super().__init__(state_dimen, action_dimen)
        if state_dimen > _lib.SX_MAX_NS or action_dimen > _lib.SX_MAX_NU:
            raise ValueError(f'state/action dimension ({state_dimen}, {action_dimen}) beyond the compiled limits')
        self._device = torch.device(get_device(conf))
        self._training_iterations = int(conf.mc_dropout_training_iterations)
        self._num_mc_samples = int(conf.mc_dropout_num_samples)
        self._predict_std = bool(conf.mc_dropout_predict_std)
        self._reinitialize_on_train = bool(conf.mc_dropout_reinitialize)
        self._hidden = [int(h) for h in conf.mc_dropout_hidden_features]
        if len(self._hidden) > _lib.SX_MLP_MAX_HIDDEN or (self._hidden and max(self._hidden) > _lib.SX_MLP_MAX_WIDTH):
            raise NotImplementedError(f'mc_dropout_hidden_features={self._hidden}: the device kernel holds up to '
                                      f'{_lib.SX_MLP_MAX_HIDDEN} hidden layers of up to {_lib.SX_MLP_MAX_WIDTH} units')
        # three and four hidden layers run on the one-particle-per-lane kernel (csrc/sx_mlp.hpp), which keeps
        # (layers + 2) x widest-layer x 64 doubles in LDS (160 KB): 3 x 64 fits, 4 layers up to 53 units
        if len(self._hidden) >= 3 and (len(self._hidden) + 2) * max(self._hidden) * 64 * 8 > 160 * 1024:
            raise NotImplementedError(f'mc_dropout_hidden_features={self._hidden}: with {len(self._hidden)} hidden layers the '
                                      f'widest may have {160 * 1024 // ((len(self._hidden) + 2) * 64 * 8)} units')
        self._type = conf.mc_dropout_type
        if self._type == 'fixed':
            self._rate = float(conf.mc_dropout_fixed_probability)
            if self._predict_std:
                raise ValueError('Predicting aleatoric uncertainty is not supported for fixed dropout.')   # reference :155
        elif self._type == 'concrete':
            self._rate = float(getattr(conf, 'mc_dropout_concrete_initial_probability', 0.1))
        else:
            raise ValueError(f'Unknown dropout type {self._type}')
        self._on_input = bool(conf.mc_dropout_on_input)
        self._seed = int(getattr(conf, 'mc_dropout_seed', 0))
        self._gen = torch.Generator(device=self._device)
        self._gen.manual_seed(self._seed)
        self._constructs = 0
        self._model = self._construct()
        self._last_training_losses: List[float] = []
        self._mlp: Optional[_lib.SxMlpModel] = None
        self._buffers = ()
        self._freeze()

    def _construct(self) -> _DropoutNet:
        # (seeded, and a different seed every time: mc_dropout_reinitialize constructs the network again before each
        # training and must not get the same weights back -- reference dropout_ssm_cem.py:115-117, test_ssm_cem.py:66-86)
        state = torch.random.get_rng_state()
        torch.manual_seed(self._seed + self._constructs)
        self._constructs += 1
        out_features = self.num_states * 2 if self._predict_std else self.num_states
        net = _DropoutNet(self.num_states + self.num_actions, out_features, self._hidden, self._rate,
                          self._type == 'concrete', self._on_input).to(torch.float64).to(self._device)
        torch.random.set_rng_state(state)
        return net

    # ---- the frozen ensemble: what the device kernels see ------------------------------------------------------
    def _freeze(self, masks=None) -> None:
        """Draws the members' masks (seeded) and lays weights and masks out for sx_mlp_predict / sx_cem_rollout_mlp.
        `masks`: the members' multipliers given instead of drawn ([S x width] per masked layer: input, hidden ...) -- how the
        golden test puts the reference's recorded noise behind the device kernels."""
        net = self._model
        with torch.no_grad():
            if masks is None:
                masks = net.draw_masks((self._num_mc_samples,), generator=self._gen)
            else:
                masks = [torch.as_tensor(m, dtype=torch.float64, device=self._device) for m in masks]
                assert [tuple(m.shape) for m in masks] == [(self._num_mc_samples, w) for w in net.sizes]
            mask_buf = torch.cat(masks, dim=1).contiguous()                                   # [S x sum widths]
            parts = []
            for lin in list(net.linears) + [net.out]:
                parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
            net_buf = torch.cat(parts).to(torch.float64).contiguous()
        m = _lib.SxMlpModel()
        m.n_s, m.n_u, m.n_hidden = self.num_states, self.num_actions, len(self._hidden)
        m.n_out, m.n_samples, m.predict_std = net.out.out_features, self._num_mc_samples, int(bool(self._predict_std))
        for i, w in enumerate(net.sizes):
            m.width[i] = w
        m.net, m.masks = net_buf.data_ptr(), mask_buf.data_ptr()
        self._mlp, self._buffers = m, (net_buf, mask_buf)

    def ensemble(self):
        """(layers [(W, b), ...] incl. the output layer, masks [S x sum widths]) as numpy: what an oracle needs."""
        net = self._model
        layers = [(lin.weight.detach().cpu().numpy(), lin.bias.detach().cpu().numpy()) for lin in list(net.linears) + [net.out]]
        return layers, self._buffers[1].cpu().numpy()

    @property
    def mlp_model(self) -> _lib.SxMlpModel:
        return self._mlp

    # ---- prediction ------------------------------------------------------------------------------------------------
    def _predict_z(self, z: Tensor, jacobians: bool):
        n, d_in = z.size(0), self.num_states + self.num_actions
        assert_shape(z, (n, d_in))
        _lib.require_gpu(z, 'states/actions')
        z = z.detach().contiguous()
        mean = torch.empty((n, self.num_states), dtype=torch.float64, device=z.device)
        var = torch.empty_like(mean)
        jac = torch.empty((n, self.num_states, d_in), dtype=torch.float64, device=z.device) if jacobians else None
        if n:
            _lib.check(_lib.lib().sx_mlp_predict(ctypes.byref(self._mlp), _lib.ptr(z), n, _lib.ptr(mean), _lib.ptr(var),
                                                 _lib.ptr(jac), _lib.stream_ptr(z.device)), 'sx_mlp_predict')
        return mean, var, jac

    def predict_with_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        return self._predict_z(self._join_states_actions(states, actions), True)

    def predict_without_jacobians(self, states: Tensor, actions: Tensor) -> Tuple[Tensor, Tensor]:
        mean, var, _ = self._predict_z(self._join_states_actions(states, actions), False)
        return mean, var

    def predict_raw(self, z: Tensor):
        """[N x (n_s + n_u)] -> (mean [N x n_s], var [N x n_s]) -- NOT transposed, as the reference's dropout SSMs
        (dropout_ssm_cem.py:96-112; its GP returns [n_s x N])."""
        mean, var, _ = self._predict_z(z, False)
        return mean, var

    def workspace(self, nbytes: int):
        return None

    # ---- training: the warm path ---------------------------------------------------------------------------------------
    def _update_model(self, x_train: Tensor, y_train: Tensor) -> None:
        pass   # the data lives in the network's weights (reference :114-116)

    def _loss(self, targets: Tensor, output: Tensor, n_data: int) -> Tensor:
        means = output[:, :self.num_states]
        reg = 1e-2 * self._model.regularization(n_data)
        if self._type == 'fixed':
            return torch.nn.functional.mse_loss(means, targets) + reg
        ll = self._gaussian_log_likelihood(targets, means, output[:, self.num_states:] if self._predict_std else None)
        return (-ll + reg).mean()

    @staticmethod
    def _gaussian_log_likelihood(targets: Tensor, pred_means: Tensor, pred_log_stds: Optional[Tensor]) -> Tensor:
        """Reference dropout_ssm_cem.py:163-173 (pinned by tests/golden/dropout_gal.npz)."""
        deltas = pred_means - targets
        if pred_log_stds is not None:
            return -((deltas / pred_log_stds.exp()) ** 2).sum(-1) * 0.5 - pred_log_stds.sum(-1) - math.log(2 * math.pi) * 0.5
        return -(deltas ** 2).sum(-1) * 0.5

    def _train_model(self, x_train: Tensor, y_train: Tensor) -> None:
        if self._reinitialize_on_train:
            self._model = self._construct()
        if y_train.dim() == 1:
            y_train = y_train.unsqueeze(1)
        x, y = x_train.detach().to(self._device, torch.float64), y_train.detach().to(self._device, torch.float64)
        net = self._model
        optimizer = torch.optim.Adam([p for p in net.parameters() if p.requires_grad])
        losses = []
        for _ in range(self._training_iterations):
            optimizer.zero_grad()
            output = net(x, net.draw_masks((x.size(0),), generator=self._gen))      # fresh masks per step (resample=True)
            loss = self._loss(y, output, x.size(0))
            loss.backward()
            optimizer.step()
            losses.append(float(loss.item()))
        self._last_training_losses = losses
        if self._training_iterations > 0 or self._reinitialize_on_train:
            self._freeze()      # (nothing trained, nothing constructed: the frozen ensemble stands)

    def collect_metrics(self) -> Dict[str, Any]:
        """The dropout rate of every dropout layer under the reference's key: `dropout_p_layer_<i>` with i the layer's index in
        bnn's module list ([input dropout,] then (linear, dropout, relu) per hidden layer) -- 1, 4, 7, ... without input
        dropout, 0, 2, 5, 8, ... with it (reference dropout_ssm_cem.py:175-181, test_ssm_cem.py:88-108)."""
        ps = self._model.rates().detach().cpu()
        shift = 1 if self._on_input else 0
        out = {'dropout_p_layer_0': float(ps[0])} if self._on_input else {}
        out.update({f'dropout_p_layer_{1 + 3 * (i - 1) + shift}': float(ps[i]) for i in range(1, len(ps))})
        return out

    @property
    def parametric(self) -> bool:
        return True
