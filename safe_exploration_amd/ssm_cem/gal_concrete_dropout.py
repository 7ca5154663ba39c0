"""Concrete-dropout state-space model (Gal, Hron, Kendall), predicted by libsxamd as an ensemble of thinned networks.

Mirrors the reference's ``GalConcreteDropoutSSM`` (``safe_exploration/ssm_cem/gal_concrete_dropout.py``): two hidden layers,
concrete dropout in front of every linear layer (input included), a mean head and a log-variance head, heteroscedastic loss
(:119-121), regularisers ``length_scale^2 / N`` and ``2 / N`` (:205-207), Adam over mini-batches of 32 for
``mc_dropout_training_iterations`` epochs (:209-226), total variance = the epistemic part only (:190-194).

**Deliberate deviation.**  The reference draws fresh concrete-dropout noise on EVERY forward pass (``torch.rand_like``,
:61-75), so two calls of ``predict_raw`` differ and the Jacobian is taken on a second, differently-seeded pass (:166-172):
there is no deterministic function to be equal to.  Here ``mc_dropout_num_samples`` noise sets are drawn once after
training (seeded) and frozen; prediction is then the deterministic ensemble of ``csrc/sx_mlp.hpp`` (mean, unbiased variance
and mean Jacobian over the members), which is what the CEM rollout kernel needs: every particle of a solve sees the same model.
"""
import math
from typing import Any, Dict, List

import numpy as np
import torch
from torch import Tensor, nn

from .. import _lib
from .dropout_ssm_cem import _EPS, _TEMPERATURE, McDropoutSSM

_BATCH_SIZE = 32


class _GalNet(nn.Module):
    def __init__(self, in_features: int, out_features: int, hidden: List[int]):
        super().__init__()
        assert len(hidden) == 2, f'We only support networks with two hidden layers, got {hidden}'
        self.linear1 = nn.Linear(in_features, hidden[0])
        self.linear2 = nn.Linear(hidden[0], hidden[1])
        self.linear3_mu = nn.Linear(hidden[1], out_features)
        self.linear3_logvar = nn.Linear(hidden[1], out_features)
        init = math.log(0.1) - math.log(0.9)                                  # init_min = init_max = 0.1 (:19-26)
        self.p_logit = nn.Parameter(torch.full((4,), init))                    # drop1, drop2, drop_mu, drop_logvar
        self.sizes = [in_features, hidden[0], hidden[1]]
        self.linears = [self.linear1, self.linear2]                            # (McDropoutSSM._freeze / ensemble read these)
        self.out = self.linear3_mu

    def rates(self) -> Tensor:
        return torch.sigmoid(self.p_logit)

    @staticmethod
    def mask_from_uniform(u: Tensor, p) -> Tensor:
        """The multiplier concrete dropout applies for uniform noise u and drop probability p (reference :49-66):
        (1 - sigmoid((logit p + logit u) / 0.1)) / (1 - p).  Pinned to the reference by tests/golden/dropout_gal.npz."""
        drop = torch.sigmoid((torch.log(p + _EPS) - torch.log(1 - p + _EPS) + torch.log(u + _EPS) - torch.log(1 - u + _EPS))
                             / _TEMPERATURE)
        return (1 - drop) / (1 - p)

    @classmethod
    def _mask(cls, shape, p, dev, generator=None) -> Tensor:
        return cls.mask_from_uniform(torch.rand(shape, dtype=torch.float64, device=dev, generator=generator), p)

    def draw_masks(self, shape_prefix, generator=None) -> List[Tensor]:
        """Multipliers of the input, of h1 and of h2 as the MEAN head sees it."""
        p, dev = self.rates(), self.p_logit.device
        return [self._mask(shape_prefix + (w,), p[i], dev, generator) for i, w in enumerate(self.sizes)]

    def forward_train(self, x: Tensor, weight_regularizer: float, dropout_regularizer: float, generator=None, uniforms=None):
        """One stochastic pass (reference _Model.forward, :95-105) -> (mean, log_var, regularisation).  `uniforms`: the four
        layers' noise [n x width] (or broadcastable) instead of fresh draws -- how the golden test replays the reference."""
        p, dev = self.rates(), x.device
        n = x.size(0)
        widths = [self.sizes[0], self.sizes[1], self.sizes[2], self.sizes[2]]
        m = [self.mask_from_uniform(uniforms[i], p[i]) if uniforms is not None else self._mask((n, widths[i]), p[i], dev, generator)
             for i in range(4)]
        a0 = x * m[0]
        h1 = torch.relu(self.linear1(a0))
        h2 = torch.relu(self.linear2(h1 * m[1]))
        mean = self.linear3_mu(h2 * m[2])
        log_var = self.linear3_logvar(h2 * m[3])
        reg = torch.zeros((), dtype=torch.float64, device=dev)
        dims = [self.sizes[0], self.sizes[1], self.sizes[2], self.sizes[2]]
        for i, lin in enumerate((self.linear1, self.linear2, self.linear3_mu, self.linear3_logvar)):
            sq = lin.weight.pow(2).sum() + lin.bias.pow(2).sum()
            reg = reg + weight_regularizer * sq / (1 - p[i]) \
                + (p[i] * torch.log(p[i]) + (1 - p[i]) * torch.log(1 - p[i])) * dropout_regularizer * dims[i]
        return mean, log_var, reg


class GalConcreteDropoutSSM(McDropoutSSM):
    def __init__(self, conf, state_dimen: int, action_dimen: int):
        assert conf.mc_dropout_on_input is True
        assert conf.mc_dropout_type == 'concrete'
        assert conf.mc_dropout_predict_std is True
        self._length_scale = float(conf.mc_dropout_lengthscale)
        super().__init__(conf, state_dimen, action_dimen)
        self._predict_std = False        # the prediction is the epistemic variance of the mean head only (:190-194)
        self._freeze()

    def _construct(self) -> _GalNet:
        state = torch.random.get_rng_state()
        torch.manual_seed(self._seed)
        net = _GalNet(self.num_states + self.num_actions, self.num_states, self._hidden).to(torch.float64).to(self._device)
        torch.random.set_rng_state(state)
        return net

    def _train_model(self, x_train: Tensor, y_train: Tensor) -> None:
        if y_train.dim() == 1:
            y_train = y_train.unsqueeze(1)
        x, y = x_train.detach().to(self._device, torch.float64), y_train.detach().to(self._device, torch.float64)
        n = x.size(0)
        weight_regularizer, dropout_regularizer = self._length_scale ** 2.0 / n, 2.0 / n
        net = self._construct()                                       # a fresh model per training, as the reference (:208)
        optimizer = torch.optim.Adam(net.parameters())
        losses = []
        for _ in range(self._training_iterations):
            for b in range(int(np.ceil(n / _BATCH_SIZE))):
                xb, yb = x[_BATCH_SIZE * b:_BATCH_SIZE * (b + 1)], y[_BATCH_SIZE * b:_BATCH_SIZE * (b + 1)]
                mean, log_var, reg = net.forward_train(xb, weight_regularizer, dropout_regularizer, self._gen)
                loss = torch.mean(torch.sum(torch.exp(-log_var) * (yb - mean) ** 2 + log_var, 1), 0) + reg   # :119-121
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                losses.append(float(loss.item()))
        self._last_training_losses = losses
        self._model = net
        self._freeze()

    def collect_metrics(self) -> Dict[str, Any]:
        ps = self._model.rates().detach().cpu()
        names = ('conc_drop1', 'conc_drop2', 'conc_drop_mu', 'conc_drop_logvar')
        return {**{'dropout_p_' + k: float(v) for k, v in zip(names, ps)}, 'losses': self._last_training_losses}
