"""GPU: the reference's numerical-failure paths through the C ABI: `_fix_zeros_nans`' whole-batch
rule, the `u_b > 0` assertion, the failure dump, and the step-by-step path `FusedCemMpc` falls back to."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import cem as ocem
from oracle import reachability as oreach
from oracle.gp import ExactGP

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


class StubGp:
    """oracle-side stand-in: returns the crafted (mean, var, jac) whatever the query."""

    def __init__(self, mean, var, jac):
        self.mean, self.var, self.jac = mean, var, jac

    def predict(self, z, jacobians=True):
        return self.mean.copy(), self.var.copy(), (self.jac.copy() if jacobians else None)


def stub_ssm(mean, var, jac, n_s, n_u):
    from safe_exploration_amd.ssm_cem.ssm_cem import CemSSM

    class Stub(CemSSM):
        def predict_with_jacobians(self, states, actions):
            return T(mean), T(var), T(jac)

        def predict_without_jacobians(self, states, actions):
            return T(mean), T(var)

        def predict_raw(self, z):
            return T(mean).t(), T(var).t()

        def _update_model(self, x, y):
            pass

        def _train_model(self, x, y):
            pass

        def collect_metrics(self):
            return {}

        @property
        def parametric(self):
            return False

    return Stub(n_s, n_u)


def crafted_batch(P=70, n_s=2, n_u=1, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.normal(0, 0.1, size=(P, n_s))
    u = rng.normal(0, 0.2, size=(P, n_u))
    a = np.eye(n_s) + 0.05 * rng.normal(size=(n_s, n_s))
    b = rng.normal(size=(n_s, n_u))
    k_fb = 0.3 * rng.normal(size=(n_u, n_s))
    m = rng.normal(size=(P, n_s, n_s))
    q = 0.01 * (m @ m.transpose(0, 2, 1)) + 1e-4 * np.eye(n_s)
    mean = rng.normal(0, 0.01, size=(P, n_s))
    var = rng.uniform(1e-4, 1e-2, size=(P, n_s))
    jac = 0.1 * rng.normal(size=(P, n_s, n_s + n_u))
    return dict(p=p, u=u, a=a, b=b, k_fb=k_fb, q=q, mean=mean, var=var, jac=jac, l_mu=np.array([.05, .02]),
                l_sigma=np.array([.05, .02]), beta=2.0)


def run_both(c, with_q):
    """(HIP result or raised exception type, oracle result or raised exception type)."""
    from safe_exploration_amd.gp_reachability_pytorch import onestep_reachability
    n_s, n_u = c['p'].shape[1], c['u'].shape[1]
    ssm = stub_ssm(c['mean'], c['var'], c['jac'], n_s, n_u)
    try:
        p1, q1, sig = onestep_reachability(T(c['p']), ssm, T(c['u']), T(c['l_mu']), T(c['l_sigma']), T(c['q']) if with_q else None,
                                           T(c['k_fb']), c['beta'], 0, T(c['a']), T(c['b']))
        hip = (p1.cpu().numpy(), q1.cpu().numpy(), sig.cpu().numpy())
    except (ValueError, AssertionError) as exc:
        hip = type(exc)
    try:
        ref = oreach.onestep_reachability(c['p'], StubGp(c['mean'], c['var'], c['jac']), c['u'], c['l_mu'], c['l_sigma'],
                                          c['q'] if with_q else None, c['k_fb'], c['beta'], a=c['a'], b=c['b'])[:3]
    except (ValueError, AssertionError) as exc:
        ref = type(exc)
    return hip, ref


@pytest.mark.parametrize('with_q', [False, True])
@pytest.mark.parametrize('case', ['clean', 'exact_zero', 'zero_and_negative', 'negative_only', 'nan', 'two_zeros_many_negatives'])
def test_fix_zeros_nans_whole_batch_rule(case, with_q, capsys):
    """gp_reachability_pytorch.py:234-243 through sx_onestep_reach: an exact zero ANYWHERE in the variance batch lifts
    every non-positive entry (also the negative ones, also in other particles) to 1e-5 and the step carries on with a
    warning; without a zero a negative variance ends in ValueError; a NaN always does."""
    c = crafted_batch()
    if case == 'exact_zero':
        c['var'][17, 1] = 0.0
    elif case == 'zero_and_negative':
        c['var'][17, 1] = 0.0
        c['var'][63, 0] = -3e-7            # another particle, another workgroup of the kernel
    elif case == 'two_zeros_many_negatives':
        c['var'][0, 0] = 0.0
        c['var'][69, 1] = -0.0
        c['var'][5:60:7, 0] = -1e-9
    elif case == 'negative_only':
        c['var'][63, 0] = -3e-7
    elif case == 'nan':
        c['var'][3, 0] = np.nan
    hip, ref = run_both(c, with_q)
    if case in ('negative_only', 'nan'):
        assert hip is ValueError and ref is ValueError
        return
    assert not isinstance(hip, type) and not isinstance(ref, type)
    for h, r in zip(hip, ref):
        np.testing.assert_allclose(h, r, rtol=1e-9, atol=1e-13)
    lifted = (c['var'] <= 0)
    if lifted.any():
        assert (hip[2][lifted] == 1e-5).all()                       # sigma out = the fixed-up variance
        assert 'found 0' in capsys.readouterr().out                 # the reference's warning
    else:
        assert np.array_equal(hip[2], c['var'])


def test_status_word_bits_and_scratch_bit_is_clear_on_return():
    from safe_exploration_amd import _lib
    from safe_exploration_amd.gp_reachability_pytorch import make_env
    c = crafted_batch(P=300)
    c['var'][200, 0] = 0.0
    c['var'][10, 1] = -1e-8
    env = make_env(2, 1, a=c['a'], b=c['b'], k_fb=c['k_fb'], l_mu=c['l_mu'], l_sigma=c['l_sigma'], beta=c['beta'])
    P = 300
    outs = [torch.empty((P, 2), dtype=torch.float64, device=DEV), torch.empty((P, 2, 2), dtype=torch.float64, device=DEV),
            torch.empty((P, 2), dtype=torch.float64, device=DEV)]
    status = torch.full((1,), 0x10000, dtype=torch.int32, device=DEV)    # a stale scratch bit must not leak in or out
    args = [T(c[k]) for k in ('p', 'q', 'u', 'mean', 'var', 'jac')]
    _lib.check(_lib.lib().sx_onestep_reach(ctypes.byref(env), P, *[_lib.ptr(t) for t in args], *[_lib.ptr(t) for t in outs],
                                           _lib.ptr(status), _lib.stream_ptr(torch.device(DEV))), 'sx_onestep_reach')
    assert int(status.item()) == _lib.SX_STATUS_ZERO_FIX
    # the same batch without the zero: the negative goes to sqrt -> NaN
    c['var'][200, 0] = 1e-3
    status.zero_()
    args = [T(c[k]) for k in ('p', 'q', 'u', 'mean', 'var', 'jac')]
    _lib.check(_lib.lib().sx_onestep_reach(ctypes.byref(env), P, *[_lib.ptr(t) for t in args], *[_lib.ptr(t) for t in outs],
                                           _lib.ptr(status), _lib.stream_ptr(torch.device(DEV))), 'sx_onestep_reach')
    assert int(status.item()) & _lib.SX_STATUS_NAN and not int(status.item()) & 0x10000


@pytest.mark.parametrize('what', ['l_mu_zero', 'beta_zero_point', 'zero_q'])
def test_nonpositive_box_bound_fails_the_assertion(what):
    """ellipsoid_from_rectangle asserts u_b > 0 (utils_ellipsoid.py:304): SX_STATUS_UB_NONPOS -> AssertionError, where
    the oracle's restatement raises the same."""
    c = crafted_batch()
    with_q = True
    if what == 'l_mu_zero':
        c['l_mu'] = np.array([0.05, 0.0])          # ub_mean = l_mu * r^2 = 0
    elif what == 'zero_q':
        c['q'][11] = 0.0                           # r^2 = 0 for one particle -> ub_mean = 0 (and c = sqrt(x / 0))
    else:
        with_q = False
        c['beta'] = -1.0                           # rkhs_bounds = beta sqrt(var) < 0
    hip, ref = run_both(c, with_q)
    assert hip is AssertionError and ref is AssertionError


def healthy_solver(E=1, iters=3, P=160, H=6, hook=False):
    from safe_exploration_amd import problems
    from safe_exploration_amd.cem_mpc import FusedCemMpc
    spec = problems.pendulum(n_train=90, seed=6, obj_mode=1)
    ssm, env = problems.build(spec, DEV)
    mpc = FusedCemMpc(ssm, env, H, P, 16, iters, device=DEV, init_std=0.2)
    rng = np.random.default_rng(9)
    noise = rng.normal(size=(iters, E, P, H, 1))
    x0 = rng.normal(0, 0.02, size=(E, 2))
    return spec, mpc, x0, noise


@pytest.mark.parametrize('E', [1, 3])
def test_stepwise_path_equals_fused_path_and_oracle(E):
    """The step-by-step solve (H x (sx_gp_predict + sx_onestep_reach) + costs per iteration -- the way the reference's
    optimiser drives its callbacks) selects the same actions as the fused kernel and as the oracle."""
    from safe_exploration_amd import problems
    spec, mpc, x0, noise = healthy_solver(E)
    fused, ok_f, _, st_f = mpc.solve(T(x0), noise=T(noise))
    step, ok_s, _, st_s = mpc.solve(T(x0), noise=T(noise), stepwise=True)
    assert int(st_f.item()) == int(st_s.item()) == 0 and torch.equal(ok_f, ok_s)
    np.testing.assert_allclose(step.cpu().numpy(), fused.cpu().numpy(), rtol=0, atol=1e-9)
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    for e in range(E):
        ref, _ = ocem.cem_solve(problems.oracle_problem(spec, ocem), gp, x0[e], noise[:, e], 16, init_std=np.full((6, 1), 0.2))
        assert (ref is not None) == bool(ok_s[e])
        if ref is not None:
            np.testing.assert_allclose(step[e].cpu().numpy(), ref, rtol=0, atol=1e-9)


def test_ambiguous_status_triggers_the_stepwise_fallback():
    """NaN and zero-fix reported together is the one case in which the fused kernel's per-particle fix-up can differ from
    the reference's whole-batch rule: get_actions then repeats the solve, same draws, through the step-by-step path."""
    from safe_exploration_amd import _lib
    spec, mpc, x0, noise = healthy_solver()
    plain, _ = mpc.get_actions(T(np.concatenate((x0[0], np.zeros(4)))[None]))   # draws its own noise; remember it
    drawn = mpc._last_noise.clone()
    calls = []
    real = mpc.solve

    def spy(x, noise=None, stepwise=False, **kw):
        calls.append(stepwise)
        best, ok, hist, status = real(x, noise=drawn if noise is None else noise, stepwise=stepwise, **kw)
        if not stepwise:
            status = status | (_lib.SX_STATUS_NAN | _lib.SX_STATUS_ZERO_FIX)     # what an ambiguous fused solve reports
        return best, ok, hist, status

    mpc.solve = spy
    again, _ = mpc.get_actions(T(np.concatenate((x0[0], np.zeros(4)))[None]))
    assert calls == [False, True] and mpc.stepwise_fallbacks == 1 and mpc.last_status == 0
    assert (plain is None) == (again is None)
    if plain is not None:
        np.testing.assert_allclose(again.cpu().numpy(), plain.cpu().numpy(), rtol=0, atol=1e-9)


def test_failure_dump_is_written_before_the_value_error(tmp_path, monkeypatch):
    """The reference saves GP state, inputs and training data to negative_variance_state.pt before it raises
    (gp_reachability_pytorch.py:256-266); so do get_actions and onestep_reachability here."""
    from safe_exploration_amd.gp_reachability_pytorch import FAILURE_DUMP_PATH, onestep_reachability
    monkeypatch.chdir(tmp_path)
    spec, mpc, x0, _ = healthy_solver()
    flat = torch.zeros((1, 6), dtype=torch.float64, device=DEV)
    flat[0, 1] = float('nan')
    with pytest.raises(ValueError):
        mpc.get_actions(flat)
    dump = torch.load(os.path.join(tmp_path, FAILURE_DUMP_PATH), weights_only=True)
    assert set(dump) == {'gp_model', 'gp_likelihood', 'state', 'action', 'x_train', 'y_train'}
    assert tuple(dump['x_train'].shape) == (90, 3) and torch.isnan(dump['state']).any()
    assert set(dump['gp_model']) == {'raw_lengthscale', 'raw_outputscale'} and 'raw_noise' in dump['gp_likelihood']
    os.remove(os.path.join(tmp_path, FAILURE_DUMP_PATH))
    ssm = mpc._ssm
    p = T([[float('nan'), 0.0]])
    with pytest.raises(ValueError):
        onestep_reachability(p, ssm, T([[0.1]]), T(spec.l_mu), T(spec.l_sigma), None, T(spec.k_fb), spec.beta, 0, T(spec.a), T(spec.b))
    assert os.path.exists(os.path.join(tmp_path, FAILURE_DUMP_PATH))
    # the state dict restores the hyper-parameters
    from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM

    class Conf:
        device, exact_gp_kernel, exact_gp_training_iterations = DEV, 'rbf', 0

    other = GpCemSSM(Conf(), 2, 1)
    other.load_state_dict(ssm.state_dict())
    assert torch.equal(other.lengthscale, ssm.lengthscale) and torch.equal(other.noise, ssm.noise)


def test_pack_result_is_the_one_hand_off_of_a_solve():
    """sx_cem_pack_result: status words, feasibility flags, the point-state check and the selected actions in one buffer
    (what get_actions / get_actions_batch copy to the host, once)."""
    import ctypes
    from safe_exploration_amd import _lib
    lib = _lib.lib()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    for G, E, L, q in [(1, 1, 15, None), (8, 1, 30, np.zeros(4)), (1, 5, 7, np.zeros((5, 16))), (2, 3, 4, 'nonzero'),
                       (1, 2, 3, 'nan')]:
        status = torch.tensor(rng.integers(0, 8, size=G), dtype=torch.int32, device=dev)
        ok = torch.tensor(rng.integers(0, 2, size=E), dtype=torch.int32, device=dev)
        best = torch.tensor(rng.normal(size=(E, L)), dtype=torch.float64, device=dev)
        if isinstance(q, str):
            qb = np.zeros(1000)
            qb[777] = np.nan if q == 'nan' else -1e-300
            want_flag = 1.0
        else:
            qb, want_flag = q, 0.0
        qt = None if qb is None else torch.tensor(np.asarray(qb).reshape(-1), dtype=torch.float64, device=dev)
        out = torch.full((G + E + 1 + E * L,), 7.0, dtype=torch.float64, device=dev)
        _lib.check(lib.sx_cem_pack_result(G, E, L, _lib.ptr(status), _lib.ptr(ok), _lib.ptr(qt), 0 if qt is None else qt.numel(),
                                          _lib.ptr(best), _lib.ptr(out), _lib.stream_ptr(dev)), 'sx_cem_pack_result')
        got = out.cpu().numpy()
        np.testing.assert_array_equal(got[:G], status.cpu().numpy().astype(np.float64))
        np.testing.assert_array_equal(got[G:G + E], ok.cpu().numpy().astype(np.float64))
        assert got[G + E] == want_flag
        np.testing.assert_array_equal(got[G + E + 1:], best.cpu().numpy().reshape(-1))
    assert lib.sx_cem_pack_result(0, 1, 1, None, None, None, 0, None, None, None) == _lib.SX_ERR_ARG
