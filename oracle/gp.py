"""Oracle: exact multi-output GP posterior (numpy, float64).  TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates what the reference asks gpytorch 0.3.2 to compute (PARITY UNPINNED on values: gpytorch is not available):

* model: ``n_s`` independent exact GPs on shared inputs, zero mean, ``ScaleKernel(RBFKernel(ard))``, Gaussian
  likelihood -- reference ``safe_exploration/ssm_cem/gp_ssm_cem.py:33-57`` and
  ``safe_exploration/ssm_pytorch/gaussian_process.py:71-79,136-140``;
* prediction INCLUDES the likelihood noise (``self._likelihood(self._model(z))``, ``gp_ssm_cem.py:89-94``);
* Jacobian of the mean w.r.t. the stacked input ``z=(x,u)`` laid out ``[N x n_s x (n_s+n_u)]``
  (``gp_ssm_cem.py:59-73``, ``ssm_pytorch/utilities.py:54-85`` get it from autograd; here it is analytic).
"""
import numpy as np
import scipy.linalg as sla


class ExactGP:
    """Hyper-parameters are explicit inputs; nothing is learned here.

    lengthscale [n_s x D], outputscale [n_s], noise [n_s]; X [N x D]; Y [N x n_s].
    """

    def __init__(self, X, Y, lengthscale, outputscale, noise):
        self.X = np.asarray(X, dtype=np.float64)
        self.Y = np.asarray(Y, dtype=np.float64)
        self.n, self.D = self.X.shape
        self.n_s = self.Y.shape[1]
        self.ls = np.broadcast_to(np.asarray(lengthscale, dtype=np.float64), (self.n_s, self.D)).copy()
        self.s = np.broadcast_to(np.asarray(outputscale, dtype=np.float64), (self.n_s,)).copy()
        self.noise = np.broadcast_to(np.asarray(noise, dtype=np.float64), (self.n_s,)).copy()
        self.L = []      # lower Cholesky factors of K_d + noise_d I
        self.alpha = []  # (K_d + noise_d I)^-1 y_d
        for d in range(self.n_s):
            K = self.kernel(d, self.X, self.X) + self.noise[d] * np.eye(self.n)
            L = np.linalg.cholesky(K)
            self.L.append(L)
            self.alpha.append(sla.cho_solve((L, True), self.Y[:, d]))

    def kernel(self, d, A, B):
        """k_d(a,b) = s_d exp(-1/2 sum_j ((a_j-b_j)/l_dj)^2)   [len(A) x len(B)]"""
        a = A / self.ls[d]
        b = B / self.ls[d]
        out = np.empty((a.shape[0], b.shape[0]))
        step = max(1, 2_000_000 // max(1, b.shape[0] * self.D))
        for i in range(0, a.shape[0], step):  # direct differences: no cancellation from the expanded square
            diff = a[i:i + step, None, :] - b[None, :, :]
            out[i:i + step] = (diff * diff).sum(2)
        return self.s[d] * np.exp(-0.5 * out)

    def predict(self, z, jacobians=True):
        """z [P x D] -> mean [P x n_s], var [P x n_s] (noise included), jac [P x n_s x D] or None."""
        z = np.asarray(z, dtype=np.float64)
        P = z.shape[0]
        mean = np.empty((P, self.n_s))
        var = np.empty((P, self.n_s))
        jac = np.empty((P, self.n_s, self.D)) if jacobians else None
        for d in range(self.n_s):
            ks = self.kernel(d, z, self.X)                     # [P x N]
            mean[:, d] = ks @ self.alpha[d]
            v = sla.solve_triangular(self.L[d], ks.T, lower=True)  # L^-1 k*   [N x P]
            var[:, d] = self.s[d] - (v * v).sum(0) + self.noise[d]
            if jacobians:
                w = ks * self.alpha[d][None, :]               # [P x N]
                # d mean_d / d z_j = sum_i alpha_i k_i (X_ij - z_j) / l_dj^2
                jac[:, d, :] = (w @ self.X - w.sum(1)[:, None] * z) / (self.ls[d] ** 2)[None, :]
        return mean, var, jac

    def variance_jacobian(self, z):
        """d var_d / d z_j = 2 sum_i v_i k*_i (z_j - X_ij) / l_dj^2 with v = (K_d + noise_d I)^-1 k*   [P x n_s x D].
        (The reference differentiates gpytorch's variance with autograd, ssm_pytorch/gaussian_process.py:222-231;
        closed form here, pinned by finite differences of `predict` in tests/test_oracle_golden.py.)"""
        z = np.asarray(z, dtype=np.float64)
        out = np.empty((z.shape[0], self.n_s, self.D))
        for d in range(self.n_s):
            ks = self.kernel(d, z, self.X)                                  # [P x N]
            v = sla.cho_solve((self.L[d], True), ks.T).T                    # [P x N]
            w = v * ks
            out[:, d, :] = 2.0 * (w.sum(1)[:, None] * z - w @ self.X) / (self.ls[d] ** 2)[None, :]
        return out

    def mean_hessian(self, z):
        """d^2 mean_d / dz dz^T = sum_i alpha_i k_i [g_i g_i^T - diag(1 / l_d^2)], g_i = (z - X_i) / l_d^2   [P x n_s x D x D].
        (The reference gets it from the `hessian` package over autograd, ssm_pytorch/gaussian_process.py:160-187; closed
        form here, pinned by finite differences of `predict`'s Jacobian in tests/test_oracle_golden.py.)"""
        z = np.asarray(z, dtype=np.float64)
        out = np.empty((z.shape[0], self.n_s, self.D, self.D))
        for d in range(self.n_s):
            w = self.kernel(d, z, self.X) * self.alpha[d][None, :]                 # [P x N]
            g = (z[:, None, :] - self.X[None, :, :]) / (self.ls[d] ** 2)[None, None, :]   # [P x N x D]
            out[:, d] = np.einsum('pi,pij,pil->pjl', w, g, g) - w.sum(1)[:, None, None] * np.diag(1.0 / self.ls[d] ** 2)[None]
        return out

    # the operands the HIP kernels consume (checked against the device-side fit in tests)
    def linv(self):
        """[n_s x N x N] inverse Cholesky factors W_d = L_d^-1 (lower triangular)."""
        eye = np.eye(self.n)
        return np.stack([sla.solve_triangular(self.L[d], eye, lower=True) for d in range(self.n_s)])


class FeatureNet:
    """The feature map of the reference's NNFeatureKernel (ssm_cem/gp_ssm_cem.py:140-185), restated in numpy: Linear,
    (ReLU, Linear)*, PReLU, then per point  phi = 2 (f - min f) / max f - 1  (the max of the UN-shifted features, as the
    reference writes it).  `layers` = [(W [out x in], b [out]), ...]; no layers = the identity (the 'linear' kernel)."""

    def __init__(self, layers=(), prelu=0.25, normalise=True):
        self.layers = [(np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64)) for W, b in layers]
        self.prelu = float(prelu)
        self.normalise = bool(normalise) and len(self.layers) > 0

    def __call__(self, z, jacobian=False):
        """z [P x D] -> phi [P x F] (and d phi / dz [P x F x D])."""
        z = np.asarray(z, dtype=np.float64)
        P, D = z.shape
        a = z
        J = np.broadcast_to(np.eye(D), (P, D, D)).copy() if jacobian else None
        for i, (W, b) in enumerate(self.layers):
            if i > 0:
                mask = (a > 0).astype(np.float64)
                a = a * mask
                if jacobian:
                    J = J * mask[:, :, None]
            a = a @ W.T + b
            if jacobian:
                J = np.einsum('ok,pkd->pod', W, J)
        if not self.layers:
            return (a, J) if jacobian else a
        slope = np.where(a > 0, 1.0, self.prelu)
        f = a * slope
        if jacobian:
            J = J * slope[:, :, None]
        if self.normalise:
            imin, imax = f.argmin(1), f.argmax(1)
            mn, mx = f[np.arange(P), imin][:, None], f[np.arange(P), imax][:, None]
            phi = 2.0 * ((f - mn) / mx) - 1.0
            if jacobian:
                Jmin, Jmax = J[np.arange(P), imin], J[np.arange(P), imax]          # [P x D]
                J = 2.0 * ((J - Jmin[:, None, :]) / mx[:, :, None] - ((f - mn) / mx ** 2)[:, :, None] * Jmax[:, None, :])
        else:
            phi = f
        return (phi, J) if jacobian else phi


class FeatureGP:
    """Exact GP with the degenerate kernel k_d(x, x') = c_d phi(x) . phi(x') (reference kernels 'linear' and 'nn':
    ScaleKernel(LinearKernel / NNFeatureKernel), ssm_cem/gp_ssm_cem.py:45-57), computed in KERNEL space -- N x N Cholesky,
    exactly like ExactGP above -- so that it checks the device's weight-space form from the other side.  PARITY
    UNPINNED on values (gpytorch absent).  c [n_s] = outputscale x variance; noise is included in the variance."""

    def __init__(self, X, Y, net: FeatureNet, c, noise):
        self.X = np.asarray(X, dtype=np.float64)
        self.Y = np.asarray(Y, dtype=np.float64)
        self.n, self.D = self.X.shape
        self.n_s = self.Y.shape[1]
        self.net = net
        self.c = np.broadcast_to(np.asarray(c, dtype=np.float64), (self.n_s,)).copy()
        self.noise = np.broadcast_to(np.asarray(noise, dtype=np.float64), (self.n_s,)).copy()
        self.Phi = net(self.X)
        self.L, self.alpha = [], []
        for d in range(self.n_s):
            K = self.c[d] * self.Phi @ self.Phi.T + self.noise[d] * np.eye(self.n)
            L = np.linalg.cholesky(K)
            self.L.append(L)
            self.alpha.append(sla.cho_solve((L, True), self.Y[:, d]))

    def predict(self, z, jacobians=True):
        z = np.asarray(z, dtype=np.float64)
        phi, J = self.net(z, jacobian=True)
        P = z.shape[0]
        mean, var = np.empty((P, self.n_s)), np.empty((P, self.n_s))
        jac = np.empty((P, self.n_s, self.D)) if jacobians else None
        for d in range(self.n_s):
            ks = self.c[d] * phi @ self.Phi.T                                  # [P x N]
            mean[:, d] = ks @ self.alpha[d]
            v = sla.solve_triangular(self.L[d], ks.T, lower=True)
            var[:, d] = self.c[d] * (phi * phi).sum(1) - (v * v).sum(0) + self.noise[d]
            if jacobians:
                w = self.c[d] * self.Phi.T @ self.alpha[d]                     # d mean / d phi   [F]
                jac[:, d, :] = np.einsum('f,pfd->pd', w, J)
        return mean, var, jac

    def mll(self):
        """Exact marginal log likelihood per output [n_s]."""
        out = np.empty(self.n_s)
        for d in range(self.n_s):
            out[d] = -0.5 * self.Y[:, d] @ self.alpha[d] - np.log(np.diag(self.L[d])).sum() - 0.5 * self.n * np.log(2 * np.pi)
        return out


def concrete_dropout_mask(u, p, eps=1e-7, temperature=0.1):
    """Multiplier of concrete dropout for uniform noise u and drop probability p (reference
    ssm_cem/gal_concrete_dropout.py:49-66): x * (1 - sigmoid((logit(p) + logit(u)) / temperature)) / (1 - p), the logits with
    the reference's eps.  Pinned to the reference by tests/golden/dropout_gal.npz."""
    u = np.asarray(u, dtype=np.float64)
    logit = np.log(p + eps) - np.log(1 - p + eps) + np.log(u + eps) - np.log(1 - u + eps)
    return (1.0 - 1.0 / (1.0 + np.exp(-logit / temperature))) / (1.0 - p)


class DropoutEnsemble:
    """MC-dropout state-space model with FROZEN masks (reference ssm_cem/dropout_ssm_cem.py, gal_concrete_dropout.py): an
    ensemble of S thinned ReLU networks.  PINNED for the concrete-dropout network: with the reference's own forward pass
    replaying recorded noise (one frozen member per pass), per-member outputs, predict_raw's mean / var(0) and the mean
    Jacobian agree with this class to 1e-12 (tests/golden/dropout_gal.npz, tests/test_oracle_golden.py).  The `bnn`-based
    McDropoutSSM stays unpinned (the package is absent).
        a_0 = m_0^s * z,   a_l = relu(W_l a_{l-1} + b_l) * m_l^s,   out^s = W_out a_L + b_out,
    mean / unbiased variance over the members of the first n_s outputs (dropout_ssm_cem.py:100-112: `preds.mean(dim=0)`,
    `preds.var(dim=0)`), analytic Jacobian of the mean.  With `predict_std` the outputs n_s .. 2 n_s - 1 are log standard
    deviations and the variance gains their mean square: the expectation, over the reference's fresh aleatoric noise
    `pred_log_stds.exp() * randn`, of the sample variance it computes (:106-109).
    layers = [(W, b), ...] hidden layers then the output layer; masks [S x (D + sum of hidden widths)]."""

    def __init__(self, layers, masks, n_s, predict_std=False):
        self.layers = [(np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64)) for W, b in layers]
        self.masks = np.asarray(masks, dtype=np.float64)
        self.n_s, self.predict_std = n_s, predict_std
        self.widths = [self.layers[0][0].shape[1]] + [W.shape[0] for W, _ in self.layers[:-1]]
        assert self.masks.shape[1] == sum(self.widths)

    def predict(self, z, jacobians=True):
        z = np.asarray(z, dtype=np.float64)
        P, D = z.shape
        S = self.masks.shape[0]
        outs = np.empty((S, P, self.layers[-1][0].shape[0]))
        jac = np.zeros((P, self.n_s, D)) if jacobians else None
        offs = np.cumsum([0] + self.widths)
        for s in range(S):
            m = [self.masks[s, offs[i]:offs[i + 1]] for i in range(len(self.widths))]
            a = z * m[0]
            pres = []
            for l, (W, b) in enumerate(self.layers[:-1]):
                pre = a @ W.T + b
                pres.append(pre)
                a = np.maximum(pre, 0.0) * m[l + 1]
            Wo, bo = self.layers[-1]
            outs[s] = a @ Wo.T + bo
            if jacobians:
                G = np.broadcast_to(Wo[:self.n_s][None], (P, self.n_s, Wo.shape[1])).copy()      # d out / d a_L
                for l in range(len(self.layers) - 2, -1, -1):
                    G = G * (m[l + 1] * (pres[l] > 0))[:, None, :]                              # -> d / d pre_l
                    G = G @ self.layers[l][0]                                                    # -> d / d a_{l-1}
                jac += G * m[0][None, None, :]
        mean = outs[:, :, :self.n_s].mean(0)
        var = outs[:, :, :self.n_s].var(0, ddof=1) if S > 1 else np.zeros_like(mean)
        if self.predict_std:
            var = var + np.exp(2.0 * outs[:, :, self.n_s:2 * self.n_s]).mean(0)
        return mean, var, (jac / S if jacobians else None)
