"""GPU: the MC-dropout state-space models (reference ssm_cem/dropout_ssm_cem.py, gal_concrete_dropout.py) as
frozen-mask ensembles against the numpy oracle (oracle.gp.DropoutEnsemble): posterior, mean Jacobian, CEM rollout, full
solve, training.  Every test runs twice: on the matrix-core kernels (csrc/sx_mlp_mfma.hpp; one or two hidden layers of up to
64 units) and with SX_MLP_PATH=valu on the one-particle-per-lane kernels (csrc/sx_mlp.hpp) that serve every other network.
`bnn` is absent and the reference samples through torch's RNG: values parity-unpinned."""
import os

import numpy as np
import pytest
import torch

from oracle import cem as ocem
from oracle.gp import DropoutEnsemble

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


@pytest.fixture(autouse=True, params=['mfma', 'valu'])
def mlp_path(request):
    """the library reads SX_MLP_PATH at every launch"""
    old = os.environ.pop('SX_MLP_PATH', None)
    if request.param == 'valu':
        os.environ['SX_MLP_PATH'] = 'valu'
    yield request.param
    os.environ.pop('SX_MLP_PATH', None)
    if old is not None:
        os.environ['SX_MLP_PATH'] = old


class Conf:
    mc_dropout_training_iterations = 0
    mc_dropout_hidden_features = [8, 6]
    mc_dropout_num_samples = 12
    mc_dropout_predict_std = False
    mc_dropout_reinitialize = False
    mc_dropout_type = 'fixed'
    mc_dropout_concrete_initial_probability = 0.1
    mc_dropout_fixed_probability = 0.1
    mc_dropout_on_input = False
    mc_dropout_lengthscale = 1e-4
    device = DEV


def conf(**kw):
    return type('C', (Conf,), kw)()


def oracle_of(ssm):
    layers, masks = ssm.ensemble()
    return DropoutEnsemble(layers, masks, ssm.num_states, predict_std=bool(ssm.mlp_model.predict_std))


CASES = [dict(), dict(mc_dropout_on_input=True, mc_dropout_fixed_probability=0.3),
         dict(mc_dropout_hidden_features=[37], mc_dropout_on_input=True, mc_dropout_predict_std=True,
              mc_dropout_type='concrete', mc_dropout_num_samples=5),                      # one layer, fewer members than waves
         dict(mc_dropout_hidden_features=[64, 17], mc_dropout_num_samples=9),
         dict(mc_dropout_type='concrete', mc_dropout_predict_std=True, mc_dropout_on_input=True),
         dict(mc_dropout_hidden_features=[64, 64], mc_dropout_num_samples=30),            # the reference's default network
         dict(mc_dropout_hidden_features=[2, 2, 10, 2], mc_dropout_on_input=True),        # test_ssm_cem.py:90-101
         dict(mc_dropout_hidden_features=[], mc_dropout_on_input=True)]


@pytest.mark.parametrize('kw', CASES)
@pytest.mark.parametrize('n_s,n_u', [(2, 1), (4, 1)])
def test_ensemble_posterior_vs_oracle(kw, n_s, n_u):
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    ssm = McDropoutSSM(conf(**kw), n_s, n_u)
    assert ssm.parametric is True and ssm.kernel_family == 'mlp'
    ref = oracle_of(ssm)
    rng = np.random.default_rng(1)
    for P in (1, 64, 150):
        z = rng.normal(0, 0.7, size=(P, n_s + n_u))
        m, v, j = ssm.predict_with_jacobians(T(z[:, :n_s]), T(z[:, n_s:]))
        mo, vo, jo = ref.predict(z)
        np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(v.cpu().numpy(), vo, rtol=1e-8, atol=1e-13)
        np.testing.assert_allclose(j.cpu().numpy(), jo, rtol=1e-9, atol=1e-12)
        m2, v2 = ssm.predict_without_jacobians(T(z[:, :n_s]), T(z[:, n_s:]))
        assert torch.equal(m2, m) and torch.equal(v2, v)
        mr, vr = ssm.predict_raw(T(z))
        assert tuple(mr.shape) == (P, n_s) and torch.equal(mr, m)        # NOT transposed (dropout_ssm_cem.py:96-112)
    assert isinstance(ssm.collect_metrics(), dict)


def test_collect_metrics_keys_are_the_references(mlp_path):   # test_ssm_cem.py:88-108, test_gal_concrete_dropout.py:18-27
    if mlp_path == 'valu':
        pytest.skip('host logic: one run covers it')
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    from safe_exploration_amd.ssm_cem.gal_concrete_dropout import GalConcreteDropoutSSM
    ssm = McDropoutSSM(conf(mc_dropout_hidden_features=[2, 2, 10, 2], mc_dropout_on_input=False), 2, 1)
    assert ssm.collect_metrics().keys() == {'dropout_p_layer_1', 'dropout_p_layer_4', 'dropout_p_layer_7', 'dropout_p_layer_10'}
    ssm = McDropoutSSM(conf(mc_dropout_hidden_features=[2, 2, 10, 2], mc_dropout_on_input=True), 2, 1)
    assert ssm.collect_metrics().keys() == {'dropout_p_layer_0', 'dropout_p_layer_2', 'dropout_p_layer_5', 'dropout_p_layer_8',
                                            'dropout_p_layer_11'}
    gal = GalConcreteDropoutSSM(conf(mc_dropout_hidden_features=[3, 2], mc_dropout_type='concrete', mc_dropout_on_input=True,
                                     mc_dropout_predict_std=True), 3, 2)
    assert len([k for k in gal.collect_metrics() if k.startswith('dropout_p')]) == 4


@pytest.mark.parametrize('reinitialize', [False, True])
def test_update_model_reinitialize_on_train(mlp_path, reinitialize):   # test_ssm_cem.py:44-86
    """With 0 training iterations update_model leaves the weights alone unless mc_dropout_reinitialize is set: then the
    network is constructed again (different weights, different predictions)."""
    if mlp_path == 'valu':
        pytest.skip('host logic: one run covers it')
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    kw = dict(mc_dropout_hidden_features=[2, 2], mc_dropout_type='concrete', mc_dropout_predict_std=True,
              mc_dropout_reinitialize=reinitialize, mc_dropout_training_iterations=0)
    states, actions = T(np.ones((3, 2))), T(np.ones((3, 1)))
    torch.manual_seed(1)
    mean1, _ = McDropoutSSM(conf(**kw), 2, 1).predict_without_jacobians(states, actions)
    torch.manual_seed(1)
    ssm = McDropoutSSM(conf(**kw), 2, 1)
    ssm.update_model(torch.empty((0, 3), dtype=torch.float64, device=DEV), torch.empty((0, 2), dtype=torch.float64, device=DEV),
                     opt_hyp=True)
    mean2, var2 = ssm.predict_without_jacobians(states, actions)
    assert tuple(mean2.shape) == (3, 2) and tuple(var2.shape) == (3, 2)
    assert torch.allclose(mean1, mean2) == (not reinitialize)


def test_rollout_solve_and_get_action_over_the_ensemble():
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc, cem_rollout
    from safe_exploration_amd.gp_reachability_pytorch import make_env
    from safe_exploration_amd.safempc_cem import CemSafeMPC, MpcResult, construct_constraints
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    spec = problems.pendulum(n_train=150, seed=2, model_error=0.02)
    ssm = McDropoutSSM(conf(mc_dropout_hidden_features=[16, 16], mc_dropout_training_iterations=150), 2, 1)
    ssm.update_model(T(spec.X), T(spec.Y), replace_old=True)            # parametric: trains (150 Adam steps)
    losses = ssm._last_training_losses
    assert len(losses) == 150 and np.mean(losses[-10:]) < np.mean(losses[:10])
    ref = oracle_of(ssm)
    env = make_env(2, 1, a=spec.a, b=spec.b, k_fb=spec.k_fb, l_mu=spec.l_mu, l_sigma=spec.l_sigma, beta=spec.beta,
                   h_mat=spec.h_mat, h_vec=spec.h_vec, u_min=spec.u_min, u_max=spec.u_max)
    prob = problems.oracle_problem(spec, ocem)
    rng = np.random.default_rng(3)
    P, H = 130, 5
    acts = rng.normal(0, 0.1, size=(P, H, 1))
    x0 = np.array([0.01, -0.01])
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    want = ocem.rollout(prob, ref, x0, acts)
    traj = r['traj'][0].cpu().numpy()
    np.testing.assert_allclose(traj[:, :, :2], want.traj_p, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(traj[:, :, 2:].reshape(P, H, 2, 2), want.traj_q, rtol=1e-7, atol=1e-13)
    np.testing.assert_allclose(r['obj_cost'][0].cpu().numpy(), want.obj_cost, rtol=1e-7, atol=1e-13)
    np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), want.con_cost)
    assert int(r['status'].item()) == 0 and want.status == 0
    iters, k = 3, 13
    noise = rng.normal(size=(iters, 1, P, H, 1))
    mpc = FusedCemMpc(ssm, env, H, P, k, iters, device=DEV, init_std=0.15)
    best, ok, _, status = mpc.solve(T(x0[None]), noise=T(noise))
    step, ok_s, _, _ = mpc.solve(T(x0[None]), noise=T(noise), stepwise=True)
    ref_best, _ = ocem.cem_solve(prob, ref, x0, noise[:, 0], k, init_std=np.full((H, 1), 0.15))
    assert int(status.item()) == 0 and torch.equal(ok, ok_s) and (ref_best is not None) == bool(ok[0])
    np.testing.assert_allclose(step.cpu().numpy(), best.cpu().numpy(), rtol=0, atol=1e-9)
    if ref_best is not None:
        np.testing.assert_allclose(best[0].cpu().numpy(), ref_best, rtol=0, atol=1e-8)

    class SolverConf(Conf):
        mpc_time_horizon, cem_num_rollouts, cem_num_elites, cem_num_iterations, cem_init_std = 4, 192, 20, 3, 0.2
        use_state_constraint, use_prior_model = True, True
        plot_cem_optimisation = plot_cem_terminal_states = False

    senv = problems.StubEnv(spec, x0)
    solver = CemSafeMPC(ssm, construct_constraints(SolverConf(), senv), senv, SolverConf(), {'lin_model': (spec.a, spec.b)},
                        wx_feedback_cost=np.diag([1.0, 2.0]), wu_feedback_cost=25.0 * np.eye(1), beta_safety=spec.beta,
                        safe_policy=lambda x: spec.k_fb @ x)
    action, result = solver.get_action(x0)
    assert action.shape == (1,) and isinstance(result, MpcResult)
    m, v = solver.ssm_predict(spec.X[:3])
    assert m.shape == (3, 2) and (v >= 0).all()


def test_gal_concrete_dropout_ssm():
    from safe_exploration_amd.ssm_cem.gal_concrete_dropout import GalConcreteDropoutSSM
    c = conf(mc_dropout_type='concrete', mc_dropout_predict_std=True, mc_dropout_on_input=True,
             mc_dropout_hidden_features=[20, 12], mc_dropout_training_iterations=3, mc_dropout_num_samples=16)
    ssm = GalConcreteDropoutSSM(c, 2, 1)
    rng = np.random.default_rng(5)
    X = rng.uniform(-0.5, 0.5, size=(70, 3))
    Y = np.stack((np.sin(2 * X[:, 0]) * 0.1, X[:, 1] * X[:, 2]), 1)
    ssm.update_model(T(X), T(Y), replace_old=True)
    metrics = ssm.collect_metrics()
    assert {'dropout_p_conc_drop1', 'dropout_p_conc_drop2', 'dropout_p_conc_drop_mu', 'dropout_p_conc_drop_logvar',
            'losses'} <= set(metrics) and len(metrics['losses']) == 3 * 3            # 3 epochs x ceil(70 / 32) batches
    ref = oracle_of(ssm)
    z = rng.normal(0, 0.4, size=(90, 3))
    m, v, j = ssm.predict_with_jacobians(T(z[:, :2]), T(z[:, 2:]))
    mo, vo, jo = ref.predict(z)
    np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v.cpu().numpy(), vo, rtol=1e-8, atol=1e-14)        # epistemic variance only (:190-194)
    np.testing.assert_allclose(j.cpu().numpy(), jo, rtol=1e-9, atol=1e-12)
    m_again, _ = ssm.predict_without_jacobians(T(z[:, :2]), T(z[:, 2:]))
    assert torch.equal(m_again, m)                                                # frozen noise: a deterministic model
    with pytest.raises(AssertionError):
        GalConcreteDropoutSSM(conf(mc_dropout_type='concrete', mc_dropout_predict_std=True, mc_dropout_on_input=False), 2, 1)
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    with pytest.raises(NotImplementedError):
        McDropoutSSM(conf(mc_dropout_hidden_features=[128, 128]), 2, 1)
    with pytest.raises(ValueError):
        McDropoutSSM(conf(mc_dropout_type='fixed', mc_dropout_predict_std=True), 2, 1)


def test_gal_device_ensemble_vs_the_reference_fixture(golden_dir):
    """tests/golden/dropout_gal.npz: weights, dropout probabilities, recorded noise and the outputs of the REFERENCE's
    GalConcreteDropoutSSM.predict_raw / predict_with_jacobians when torch.rand_like replays that noise (one frozen member per
    pass; tests/golden/make_golden.py).  The device ensemble on the same weights, with its members' masks computed from the
    same noise, must return the reference's mean, var(0) and mean Jacobian."""
    from safe_exploration_amd.ssm_cem.gal_concrete_dropout import GalConcreteDropoutSSM, _GalNet
    g = np.load(os.path.join(golden_dir, 'dropout_gal.npz'))
    S = g['u_in'].shape[0]
    c = conf(mc_dropout_type='concrete', mc_dropout_predict_std=True, mc_dropout_on_input=True,
             mc_dropout_hidden_features=[int(h) for h in g['hidden']], mc_dropout_num_samples=S)
    ssm = GalConcreteDropoutSSM(c, 2, 1)
    net = ssm._model
    with torch.no_grad():
        for lin, name in ((net.linear1, '1'), (net.linear2, '2'), (net.linear3_mu, 'mu'), (net.linear3_logvar, 'logvar')):
            lin.weight.copy_(T(g['W' + name]))
            lin.bias.copy_(T(g['b' + name]))
        net.p_logit.copy_(T(np.log(g['probs']) - np.log(1 - g['probs'])))
    p = net.rates()
    ssm._freeze(masks=[_GalNet.mask_from_uniform(T(g[k]), p[i]) for i, k in enumerate(('u_in', 'u_h1', 'u_h2_mu'))])
    x = g['x']
    m, v, j = ssm.predict_with_jacobians(T(x[:, :2]), T(x[:, 2:]))
    np.testing.assert_allclose(m.cpu().numpy(), g['pred_mean'], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(v.cpu().numpy(), g['pred_var'], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(j.cpu().numpy(), g['pred_jac'], rtol=1e-10, atol=1e-12)


def test_matrix_core_kernel_equals_lane_kernel_at_the_default_size(mlp_path):
    """64 x 64, 30 members (experiments/sacred_helper.py:100-102), config 2's particle count: the two kernels against each
    other, rollout costs and trajectories (different summation orders: tolerance, not bits)."""
    if mlp_path == 'valu':
        pytest.skip('one comparison covers both')
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import cem_rollout
    from safe_exploration_amd.gp_reachability_pytorch import make_env
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    spec = problems.pendulum(n_train=100, seed=4, model_error=0.02)
    ssm = McDropoutSSM(conf(mc_dropout_hidden_features=[64, 64], mc_dropout_num_samples=30, mc_dropout_on_input=True,
                            mc_dropout_training_iterations=30), 2, 1)
    ssm.update_model(T(spec.X), T(spec.Y), replace_old=True)
    env = make_env(2, 1, a=spec.a, b=spec.b, k_fb=spec.k_fb, l_mu=spec.l_mu, l_sigma=spec.l_sigma, beta=spec.beta,
                   h_mat=spec.h_mat, h_vec=spec.h_vec, u_min=spec.u_min, u_max=spec.u_max)
    P, H = 4096 + 7, 6
    gen = torch.Generator(device=DEV).manual_seed(11)
    noise = torch.randn((1, P, H, 1), dtype=torch.float64, device=DEV, generator=gen)
    mean = torch.zeros((1, H, 1), dtype=torch.float64, device=DEV)
    std = torch.full((1, H, 1), 0.2, dtype=torch.float64, device=DEV)
    x0 = T([[0.02, -0.03]])
    out = {}
    for path in ('mfma', 'valu'):
        if path == 'valu':
            os.environ['SX_MLP_PATH'] = 'valu'
        r = cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise, want_traj=True, want_sigma=True)
        out[path] = {k: r[k].cpu().numpy() for k in ('traj', 'sigma', 'obj_cost', 'con_cost', 'actions')}
        assert int(r['status'].item()) == 0
    np.testing.assert_array_equal(out['mfma']['actions'], out['valu']['actions'])
    np.testing.assert_allclose(out['mfma']['traj'], out['valu']['traj'], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(out['mfma']['sigma'], out['valu']['sigma'], rtol=1e-8, atol=1e-16)
    np.testing.assert_allclose(out['mfma']['obj_cost'], out['valu']['obj_cost'], rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(out['mfma']['con_cost'], out['valu']['con_cost'])


@pytest.mark.parametrize('n_s,hidden,kw', [(2, [64, 64], dict(mc_dropout_on_input=True)),
                                         (4, [48], dict(mc_dropout_type='concrete', mc_dropout_predict_std=True,
                                                        mc_dropout_on_input=True)),
                                         (4, [64, 64], dict())])
def test_tiles_straddling_episodes_and_initial_ellipsoids(mlp_path, n_s, hidden, kw):
    """E = 3 episodes of 37 particles: 16-particle tiles of the matrix-core kernel hold particles of two episodes (own start
    state, own sampling distribution per lane); q0 given, so step 0 already takes the ellipsoid branch with its Jacobian.
    Both kernels against each other; the lane kernel is pinned to the oracle by the tests above."""
    if mlp_path == 'valu':
        pytest.skip('one comparison covers both')
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import cem_rollout
    from safe_exploration_amd.gp_reachability_pytorch import make_env
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    spec = problems.pendulum(n_train=50, seed=4, model_error=0.02) if n_s == 2 else problems.cartpole(n_train=50, seed=4)
    ssm = McDropoutSSM(conf(mc_dropout_hidden_features=hidden, mc_dropout_num_samples=11, **kw), n_s, 1)
    env = make_env(n_s, 1, a=spec.a, b=spec.b, k_fb=spec.k_fb, l_mu=spec.l_mu, l_sigma=spec.l_sigma, beta=spec.beta,
                   h_mat=spec.h_mat, h_vec=spec.h_vec, u_min=spec.u_min, u_max=spec.u_max)
    E, P, H = 3, 37, 4
    rng = np.random.default_rng(8)
    x0 = T(rng.normal(0, 0.02, size=(E, n_s)))
    q0 = T(np.stack([np.eye(n_s) * 1e-4 * (e + 1) for e in range(E)]))
    mean = T(rng.normal(0, 0.05, size=(E, H, 1)))
    std = T(rng.uniform(0.05, 0.2, size=(E, H, 1)))
    noise = T(rng.normal(size=(E, P, H, 1)))
    out = {}
    for path in ('mfma', 'valu'):
        if path == 'valu':
            os.environ['SX_MLP_PATH'] = 'valu'
        r = cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise, q0=q0, want_traj=True, want_sigma=True)
        out[path] = {k: r[k].cpu().numpy() for k in ('traj', 'sigma', 'obj_cost', 'con_cost', 'actions')}
        out[path]['status'] = int(r['status'].item())
    assert out['mfma']['status'] == out['valu']['status']
    np.testing.assert_array_equal(out['mfma']['actions'], out['valu']['actions'])
    np.testing.assert_allclose(out['mfma']['traj'], out['valu']['traj'], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out['mfma']['sigma'], out['valu']['sigma'], rtol=1e-8, atol=1e-18)
    np.testing.assert_allclose(out['mfma']['obj_cost'], out['valu']['obj_cost'], rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(out['mfma']['con_cost'], out['valu']['con_cost'])


def test_deep_networks_beyond_the_lane_kernels_lds_are_refused_at_construction():
    """Three hidden layers run up to 64 units wide, four up to 53 (the one-particle-per-lane kernel keeps (layers + 2) x
    widest x 64 doubles in LDS): anything beyond is refused when the model is built, not at the first prediction."""
    from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM
    for hidden in ([64, 64, 64], [53, 20, 53, 8]):
        ssm = McDropoutSSM(conf(mc_dropout_hidden_features=hidden, mc_dropout_num_samples=3), 2, 1)
        z = np.random.default_rng(0).normal(0, 0.5, size=(5, 3))
        m, v, j = ssm.predict_with_jacobians(T(z[:, :2]), T(z[:, 2:]))
        mo, vo, jo = oracle_of(ssm).predict(z)
        np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(j.cpu().numpy(), jo, rtol=1e-9, atol=1e-12)
    for hidden in ([64, 8, 8, 8], [54, 54, 54, 54], [65], [8, 8, 8, 8, 8]):
        with pytest.raises(NotImplementedError):
            McDropoutSSM(conf(mc_dropout_hidden_features=hidden), 2, 1)
