"""Generates tests/golden/*.npz by RUNNING THE REFERENCE in the build container.  Not run by the test-suite.

    python tests/golden/make_golden.py            # needs /root/reference; writes the .npz files next to this script

What runs from the reference, unmodified (imported from /root/reference, never copied):
    safe_exploration/gp_reachability_pytorch.py   onestep_reachability, lin_ellipsoid_safety_distance,
                                                  is_ellipsoid_inside_polytope
    safe_exploration/gp_reachability.py           onestep_reachability (the reference's own numpy twin, as a cross-check)
    safe_exploration/utils.py                     compute_remainder_overapproximations_pytorch
    safe_exploration/utils_ellipsoid.py           sum_two_ellipsoids_pytorch, ellipsoid_from_rectangle_pytorch
    safe_exploration/ssm_cem/gal_concrete_dropout.py   _Model.forward (:95-105, with _ConcreteDropout :28-66), _heteroscedastic_loss
                                                  (:119-121), GalConcreteDropoutSSM.predict_raw / predict_with_jacobians
                                                  (:166-196) -- with torch.rand_like replaying RECORDED uniforms that are
                                                  constant over the batch rows, so that a forward pass is one frozen
                                                  ensemble member (dropout_gal.npz)
    safe_exploration/ssm_cem/dropout_ssm_cem.py   McDropoutSSM._gaussian_log_likelihood (:163-173; static)

To import those files, modules that are not installed here (casadi, gpytorch, hessian) are replaced by NAME-ONLY
placeholders (no arithmetic; none of the functions above call into them) and ``torch.eig`` -- removed from current
torch, used at utils.py:663 -- is mapped onto ``torch.linalg.eigvals``.

The GP behind ``ssm`` is a stand-in ``CemSSM`` (gpytorch is absent): the closed-form exact GP of ``oracle/gp.py`` with
its hyper-parameters stored in the fixture.  So the fixtures pin the REACHABILITY arithmetic (p, Q, sigma passthrough,
polytope distances) to the reference; the GP values themselves stay "parity unpinned" (DESIGN.md).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def _install_placeholders():
    class _Name:
        def __init__(self, *a, **k):
            pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod('casadi', reshape=None, Callback=_Name)
    g = mod('gpytorch', ExactMarginalLogLikelihood=_Name)
    subs = dict(kernels=('ScaleKernel', 'RBFKernel', 'LinearKernel', 'Kernel'),
                likelihoods=('GaussianLikelihood', 'Likelihood'), distributions=('MultivariateNormal',),
                means=('Mean',), mlls=('MarginalLogLikelihood',), models=('ExactGP',))
    for sub, names in subs.items():
        setattr(g, sub, mod('gpytorch.' + sub, **{n: _Name for n in names}))
    mod('hessian', hessian=None)
    mod('bnn', BDropout=_Name, CDropout=_Name, bayesian_model=None)   # (dropout_ssm_cem.py:4,7; only its static loss is run)
    # torch.eig survives only as a stub that raises; give it back its old (values [n x 2], None) return
    torch.eig = lambda A, eigenvectors=False: (torch.view_as_real(torch.linalg.eigvals(A)).to(A.dtype), None)


def main():
    torch.set_default_dtype(torch.double)  # the reference runs in float64 (experiments/run.py:96-99)
    _install_placeholders()
    sys.path.insert(0, REF)
    import safe_exploration.gp_reachability_pytorch as reach_pt
    import safe_exploration.gp_reachability as reach_np
    from safe_exploration import utils as ref_utils
    from safe_exploration import utils_ellipsoid as ref_ell
    from safe_exploration.ssm_cem.ssm_cem import CemSSM
    from safe_exploration.state_space_models import StateSpaceModel

    from oracle.gp import ExactGP

    class StandInSSM(CemSSM):
        """CemSSM surface over the closed-form GP (gpytorch is absent)."""

        def __init__(self, gp, n_s, n_u):
            super().__init__(n_s, n_u)
            self.gp = gp

        def predict_with_jacobians(self, states, actions):
            z = torch.cat((states, actions), dim=1).detach().numpy()
            m, v, j = self.gp.predict(z, True)
            return torch.tensor(m), torch.tensor(v), torch.tensor(j)

        def predict_without_jacobians(self, states, actions):
            z = torch.cat((states, actions), dim=1).detach().numpy()
            m, v, _ = self.gp.predict(z, False)
            return torch.tensor(m), torch.tensor(v)

        def predict_raw(self, z):
            m, v, _ = self.gp.predict(z.detach().numpy(), False)
            return torch.tensor(m.T), torch.tensor(v.T)

        def _update_model(self, x, y):
            pass

        def _train_model(self, x, y):
            pass

        def collect_metrics(self):
            return {}

        @property
        def parametric(self):
            return False

    class NumpySSM(StateSpaceModel):
        """Single-item numpy adapter, the shape conventions of test_gp_reachability_pytorch.py:21-69."""

        def __init__(self, gp, n_s, n_u):
            super().__init__(n_s, n_u)
            self.gp = gp

        def predict(self, states, actions, jacobians=False, full_cov=False):
            z = np.concatenate((states, actions), axis=1)
            m, v, j = self.gp.predict(z, jacobians)
            if jacobians:
                return m.T, v[0], j[0]
            return m.T, v[0]

        def linearize_predict(self, *a, **k):
            raise NotImplementedError

        def get_reverse(self, seed):
            raise NotImplementedError

        def get_linearize_reverse(self, seed):
            raise NotImplementedError

        def update_model(self, *a, **k):
            raise NotImplementedError

    ref_test = os.path.join(REF, 'safe_exploration', 'test')

    def onestep_case(name, X, Y, n_s, n_u, ls, s, noise, a, b, P, seed, l_val, c_safety, q_scale, steps=3):
        """Point branch, ellipsoid branch and a chained `steps`-step rollout through the reference."""
        rng = np.random.default_rng(seed)
        gp = ExactGP(X, Y, ls, s, noise)
        ssm = StandInSSM(gp, n_s, n_u)
        l_mu = np.full(n_s, l_val)
        l_sigma = np.full(n_s, l_val)
        k_fb = rng.uniform(0, 1, size=(n_u, n_s))
        p = 0.1 * rng.uniform(0, 1, size=(P, n_s))
        k_ff = rng.uniform(0, 1, size=(P, n_u)) if steps else None
        actions = rng.uniform(-0.5, 0.5, size=(P, steps, n_u))
        actions[:, 0, :] = k_ff
        m = rng.normal(size=(P, n_s, n_s))
        q = q_scale * (m @ m.transpose(0, 2, 1) + 0.5 * np.eye(n_s)[None])
        T = torch.tensor
        ta = None if a is None else T(a)
        tb = None if b is None else T(b)
        out = dict(X=X, Y=Y, ls=gp.ls, s=gp.s, noise=gp.noise, l_mu=l_mu, l_sigma=l_sigma, k_fb=k_fb, p=p, q=q,
                   k_ff=k_ff, actions=actions, c_safety=np.float64(c_safety), has_lin=np.bool_(a is not None))
        if a is not None:
            out['a'], out['b'] = a, b
        # point branch
        p1, q1, sig = reach_pt.onestep_reachability(T(p), ssm, T(k_ff), T(l_mu), T(l_sigma), None, T(k_fb), c_safety,
                                                    verbose=0, a=ta, b=tb)
        out.update(point_p=p1.numpy(), point_q=q1.numpy(), point_sigma=sig.numpy())
        # ellipsoid branch
        p1, q1, sig = reach_pt.onestep_reachability(T(p), ssm, T(k_ff), T(l_mu), T(l_sigma), T(q), T(k_fb), c_safety,
                                                    verbose=0, a=ta, b=tb)
        out.update(ell_p=p1.numpy(), ell_q=q1.numpy(), ell_sigma=sig.numpy())
        # the reference's own numpy twin must agree (test_gp_reachability_pytorch.py:105-136)
        nssm = NumpySSM(gp, n_s, n_u)
        for i in range(P):
            pn, qn = reach_np.onestep_reachability(p[i][:, None], nssm, k_ff[i][None, :], l_mu, l_sigma, q[i], k_fb,
                                                   c_safety, verbose=0, a=a, b=b)
            assert np.allclose(pn.squeeze(), out['ell_p'][i]) and np.allclose(qn, out['ell_q'][i]), name
        # chained rollout from a point start
        pc, qc = T(p), None
        tp, tq, ts = [], [], []
        for t in range(steps):
            pc, qc, sg = reach_pt.onestep_reachability(pc, ssm, T(actions[:, t]), T(l_mu), T(l_sigma), qc, T(k_fb),
                                                       c_safety, verbose=0, a=ta, b=tb)
            tp.append(pc.numpy().copy()); tq.append(qc.numpy().copy()); ts.append(sg.numpy().copy())
        out.update(chain_p=np.stack(tp, 1), chain_q=np.stack(tq, 1), chain_sigma=np.stack(ts, 1))
        np.savez(os.path.join(HERE, name + '.npz'), **out)
        print('wrote', name)

    # (1)+(2) pendulum, training set of the reference's own fixture (test_gp_reachability_pytorch.py:74-102)
    d = np.load(os.path.join(ref_test, 'invpend_data.npz'))
    Xp, Yp = d['X'], d['y']
    rng = np.random.default_rng(125)
    a_p, b_p = rng.uniform(0, 1, size=(2, 2)), rng.uniform(0, 1, size=(2, 1))
    ls_p = np.array([[0.9, 1.3, 2.0], [1.1, 0.8, 1.7]])
    onestep_case('onestep_pendulum_lin', Xp, Yp, 2, 1, ls_p, [0.6, 0.4], [1e-2, 2e-2], a_p, b_p, 6, 1, 0.001, 2.0, 0.02)
    onestep_case('onestep_pendulum_nolin', Xp, Yp, 2, 1, ls_p, [0.6, 0.4], [1e-2, 2e-2], None, None, 6, 2, 0.001, 2.0,
                 0.02)
    # pendulum with the environment's real Lipschitz constants and beta (environments.py:476-482, sacred cem_beta_safety)
    onestep_case('onestep_pendulum_env', Xp, Yp, 2, 1, 0.7, 0.5, 1e-2, a_p * 0.5, b_p * 0.1, 5, 3, 0.05, 3.0, 0.005)

    # (7) cart-pole sized (n_s = 4) with the reference's fixture X, y, a, b (test_safempc.py:57-70)
    d = np.load(os.path.join(ref_test, 'data_cartpole.npz'))
    Xc, Yc, a_c, b_c = d['X'], d['y'], d['a'], d['b']
    ls_c = np.array([[2.0, 3.0, 1.5, 2.5, 4.0], [1.8, 2.2, 2.6, 1.4, 3.0], [2.4, 1.6, 2.0, 3.0, 2.0],
                     [3.0, 2.0, 1.2, 2.2, 2.8]])
    onestep_case('onestep_cartpole_lin', Xc, Yc, 4, 1, ls_c, [0.5, 0.8, 0.3, 0.6], [1e-2, 5e-3, 2e-2, 1e-2], a_c, b_c,
                 5, 4, 0.001, 2.0, 0.01)
    onestep_case('onestep_cartpole_nolin', Xc, Yc, 4, 1, ls_c, [0.5, 0.8, 0.3, 0.6], [1e-2, 5e-3, 2e-2, 1e-2], None,
                 None, 5, 5, 0.05, 3.0, 0.01)

    # (3) polytope distance table + inside / partial / outside (test_gp_reachability_pytorch.py:162-219)
    T = torch.tensor
    box_A = np.array([[1., 0.], [0., 1.], [-1., 0.], [0., -1.]])   # polytope.box2poly([[0,10],[0,10]]) (A x <= b)
    box_b = np.array([[10.], [10.], [0.], [0.]])
    p = np.array([[0., 0.], [1., 3.], [5., 10.]])
    q = .2 * np.array([[[.6, .21], [.21, .55]], [[.5, .2], [.2, .65]], [[.7, .21], [.21, .59]]])
    dist = reach_pt.lin_ellipsoid_safety_distance(T(p), T(q), T(box_A), T(box_b)).numpy()
    for i in range(3):
        dn = reach_np.lin_ellipsoid_safety_distance(p[i][:, None], q[i], box_A, box_b)
        assert np.allclose(dn.squeeze(1), dist[i])
    p3 = np.array([[5., 5.], [0., 0.], [20., 20.]])
    q3 = np.tile(np.array([[2., 1.], [1., 2.]]), (3, 1, 1))
    inside = reach_pt.is_ellipsoid_inside_polytope(T(p3), T(q3), T(box_A), T(box_b)).numpy()
    assert list(inside) == [True, False, False]
    rng = np.random.default_rng(7)
    pr = rng.normal(size=(16, 4))
    mm = rng.normal(size=(16, 4, 4))
    qr = mm @ mm.transpose(0, 2, 1) * 0.05
    hr = rng.normal(size=(9, 4))
    hv = rng.uniform(0.5, 2.0, size=(9, 1))
    dist_r = reach_pt.lin_ellipsoid_safety_distance(T(pr), T(qr), T(hr), T(hv)).numpy()
    inside_r = reach_pt.is_ellipsoid_inside_polytope(T(pr), T(qr), T(hr), T(hv)).numpy()
    np.savez(os.path.join(HERE, 'polytope.npz'), box_A=box_A, box_b=box_b, p=p, q=q, dist=dist, p3=p3, q3=q3,
             inside=inside, pr=pr, qr=qr, hr=hr, hv=hv, dist_r=dist_r, inside_r=inside_r)
    print('wrote polytope')

    # (4) helper tables (test_utils.py:35-59, test_utils_ellipsoid.py:19-50,112-125)
    rng = np.random.default_rng(11)
    out = {}
    for n_s, n_u in ((2, 1), (4, 1), (4, 2)):
        P = 8
        mm = rng.normal(size=(P, n_s, n_s))
        qq = mm @ mm.transpose(0, 2, 1) * 0.1
        kfb = rng.normal(size=(n_u, n_s))
        lmu = rng.uniform(0.01, 0.1, size=n_s)
        lsg = rng.uniform(0.01, 0.1, size=n_s)
        um, us = ref_utils.compute_remainder_overapproximations_pytorch(T(qq), T(kfb).repeat((P, 1, 1)),
                                                                       T(lmu).repeat((P, 1)), T(lsg).repeat((P, 1)))
        tag = f'{n_s}{n_u}'
        out.update({f'rem_q_{tag}': qq, f'rem_kfb_{tag}': kfb, f'rem_lmu_{tag}': lmu, f'rem_lsg_{tag}': lsg,
                    f'rem_umu_{tag}': um.numpy(), f'rem_usig_{tag}': us.numpy()})
        m2 = rng.normal(size=(P, n_s, n_s))
        q2 = m2 @ m2.transpose(0, 2, 1) * 0.3
        p1 = rng.normal(size=(P, n_s))
        p2 = rng.normal(size=(P, n_s))
        ps, qs = ref_ell.sum_two_ellipsoids_pytorch(T(p1), T(qq), T(p2), T(q2))
        out.update({f'sum_p1_{tag}': p1, f'sum_q1_{tag}': qq, f'sum_p2_{tag}': p2, f'sum_q2_{tag}': q2,
                    f'sum_p_{tag}': ps.numpy(), f'sum_q_{tag}': qs.numpy()})
        ub = rng.uniform(0.1, 2.0, size=(P, n_s))
        out.update({f'rect_ub_{tag}': ub, f'rect_q_{tag}': ref_ell.ellipsoid_from_rectangle_pytorch(T(ub)).numpy()})
    np.savez(os.path.join(HERE, 'helpers.npz'), **out)
    print('wrote helpers')


def dropout_main():
    """dropout_gal.npz: the reference's concrete-dropout network as a frozen ensemble (missing #1)."""
    torch.set_default_dtype(torch.double)
    _install_placeholders()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import types as _types

    import safe_exploration.ssm_cem.gal_concrete_dropout as gal
    from safe_exploration.ssm_cem.dropout_ssm_cem import McDropoutSSM as RefMcDropout

    rng = np.random.default_rng(2024)
    n_s, n_u, hidden, S, B = 2, 1, [8, 6], 5, 7
    d_in = n_s + n_u
    wr, dr = 0.7 ** 2 / 40.0, 2.0 / 40.0          # length_scale^2 / N, 2 / N (:205-207) for N = 40
    torch.manual_seed(7)
    model = gal._Model(d_in, n_s, hidden, wr, dr)
    # four different dropout probabilities (the constructor starts all of them at 0.1)
    probs = np.array([0.08, 0.15, 0.22, 0.3])
    with torch.no_grad():
        for mod_, pr in zip((model.conc_drop1, model.conc_drop2, model.conc_drop_mu, model.conc_drop_logvar), probs):
            mod_.p_logit.fill_(float(np.log(pr) - np.log(1 - pr)))
    widths = [d_in, hidden[0], hidden[1], hidden[1]]           # what the four concrete-dropout layers see
    uniforms = [rng.uniform(0.02, 0.98, size=(S, w)) for w in widths]

    # torch.rand_like, as gal_concrete_dropout.py:61 calls it: the recorded uniforms of member `member`, the same in every
    # batch row; the four layers are visited in the order conc_drop1, conc_drop2, conc_drop_mu, conc_drop_logvar (:98-103)
    state = {'member': 0, 'layer': 0}
    real_rand_like = torch.rand_like

    def replay(x, *a, **k):
        u = torch.tensor(uniforms[state['layer']][state['member']])
        assert x.shape[-1] == u.numel()
        state['layer'] += 1
        if state['layer'] == 4:
            state['layer'] = 0
            state['member'] = (state['member'] + 1) % S
        return u.expand_as(x).clone()

    x = torch.tensor(rng.uniform(-0.6, 0.6, size=(B, d_in)))
    y = torch.tensor(rng.normal(0, 0.3, size=(B, n_s)))
    out = {'n_s': n_s, 'n_u': n_u, 'hidden': np.array(hidden), 'probs': probs, 'weight_regularizer': wr,
           'dropout_regularizer': dr, 'x': x.numpy(), 'y': y.numpy(),
           'u_in': uniforms[0], 'u_h1': uniforms[1], 'u_h2_mu': uniforms[2], 'u_h2_logvar': uniforms[3]}
    for name, lin in (('1', model.linear1), ('2', model.linear2), ('mu', model.linear3_mu), ('logvar', model.linear3_logvar)):
        out['W' + name] = lin.weight.detach().numpy().copy()
        out['b' + name] = lin.bias.detach().numpy().copy()
    torch.rand_like = replay
    try:
        means, logvars, regs, losses = [], [], [], []
        for s in range(S):
            state.update(member=s, layer=0)
            mean, log_var, reg = model(x)                                    # _Model.forward :95-105
            means.append(mean.detach().numpy().copy())
            logvars.append(log_var.detach().numpy().copy())
            regs.append(float(reg))
            losses.append(float(gal._heteroscedastic_loss(y, mean, log_var)))   # :119-121
        out.update(member_mean=np.stack(means), member_logvar=np.stack(logvars), regularization=np.array(regs),
                   heteroscedastic_loss=np.array(losses))
        conf = _types.SimpleNamespace(mc_dropout_on_input=True, mc_dropout_type='concrete', mc_dropout_predict_std=True,
                                      mc_dropout_num_samples=S, mc_dropout_training_iterations=0,
                                      mc_dropout_hidden_features=hidden, mc_dropout_lengthscale=0.7, device='cpu')
        ssm = gal.GalConcreteDropoutSSM(conf, n_s, n_u)
        ssm._model = model
        state.update(member=0, layer=0)
        pm, pv = ssm.predict_raw(x)                                          # :185-196: mean, var(0) over the S passes
        state.update(member=0, layer=0)
        pm2, pv2, jac = ssm.predict_with_jacobians(x[:, :n_s].clone(), x[:, n_s:].clone())   # :166-177
        out.update(pred_mean=pm.detach().numpy(), pred_var=pv.detach().numpy(), pred_jac=jac.detach().numpy())
        assert np.allclose(pm2.detach().numpy(), out['pred_mean']) and np.allclose(pv2.detach().numpy(), out['pred_var'])
    finally:
        torch.rand_like = real_rand_like
    # McDropoutSSM's static Gaussian log likelihood (dropout_ssm_cem.py:163-173), with and without predicted log stds
    t = torch.tensor(rng.normal(size=(B, n_s)))
    pmn = torch.tensor(rng.normal(size=(B, n_s)))
    pls = torch.tensor(rng.normal(0, 0.4, size=(B, n_s)))
    out.update(ll_targets=t.numpy(), ll_means=pmn.numpy(), ll_log_stds=pls.numpy(),
               ll_with_std=RefMcDropout._gaussian_log_likelihood(t, pmn, pls).numpy(),
               ll_without_std=RefMcDropout._gaussian_log_likelihood(t, pmn, None).numpy())
    np.savez(os.path.join(HERE, 'dropout_gal.npz'), **out)
    print('wrote dropout_gal')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'dropout':
        dropout_main()
    else:
        main()
        dropout_main()
