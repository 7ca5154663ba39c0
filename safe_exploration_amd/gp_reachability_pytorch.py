"""Ellipsoidal one-step reachability and the ellipsoid-in-polytope test, evaluated by libsxamd on the GPU.

Same call signatures and return values as the reference's ``safe_exploration/gp_reachability_pytorch.py`` (:18-231).
The batched small-matrix algebra (Lagrange-remainder boxes with their largest-eigenvalue solve, box -> ellipsoid,
two ellipsoid sums) runs one particle per lane in ``sx_onestep_reach``; the GP behind ``ssm`` is whatever ``CemSSM``
the caller passes (``GpCemSSM`` -> ``sx_gp_predict``).

Error behaviour of the reference is kept: a NaN at one of its three zero/NaN checks raises ``ValueError``; an exact zero
is lifted to 1e-5 with a warning; a non-positive box bound fails the reference's assertion (``AssertionError`` here too).
Reading the status word is the one host synchronisation of a call (the reference synchronises several times).
"""
import ctypes
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .ssm_cem.ssm_cem import CemSSM
from .utils import assert_shape


def make_env(n_s: int, n_u: int, *, a=None, b=None, k_fb=None, l_mu=None, l_sigma=None, beta: float = 1.0,
             h_mat=None, h_vec=None, u_min=None, u_max=None, obj_mode: int = _lib.SX_OBJ_NEG_VARIANCE,
             obj_w_abs=None, obj_target=None, obj_w_lin=None, con_mode: int = _lib.SX_CON_ALL_STATES) -> _lib.SxEnv:
    """Host-side sx_env from numpy-convertible constants; anything omitted is zero (a omitted = identity)."""
    if n_s > _lib.SX_MAX_NS or n_u > _lib.SX_MAX_NU:
        raise ValueError(f'state/action dimension ({n_s}, {n_u}) beyond the compiled limits')
    env = _lib.SxEnv()
    env.n_s, env.n_u, env.obj_mode, env.con_mode, env.beta = n_s, n_u, obj_mode, con_mode, float(beta)
    _lib.fill(env.a, np.eye(n_s) if a is None else np.asarray(a).reshape(n_s, n_s))
    if b is not None:
        _lib.fill(env.b, np.asarray(b).reshape(n_s, n_u))
    if k_fb is not None:
        _lib.fill(env.k_fb, np.asarray(k_fb).reshape(n_u, n_s))
    if l_mu is not None:
        _lib.fill(env.l_mu, np.asarray(l_mu).reshape(n_s))
    if l_sigma is not None:
        _lib.fill(env.l_sigma, np.asarray(l_sigma).reshape(n_s))
    if h_mat is not None:
        h_mat = np.asarray(h_mat, dtype=np.float64)
        m = h_mat.shape[0]
        if m > _lib.SX_MAX_M:
            raise ValueError(f'{m} polytope rows exceed SX_MAX_M={_lib.SX_MAX_M}')
        env.m = m
        _lib.fill(env.h_mat, h_mat.reshape(m, n_s))
        _lib.fill(env.h_vec, np.asarray(h_vec).reshape(m))
    if u_min is not None:
        _lib.fill(env.u_min, np.asarray(u_min).reshape(n_u))
        _lib.fill(env.u_max, np.asarray(u_max).reshape(n_u))
    for field, val in (('obj_w_abs', obj_w_abs), ('obj_target', obj_target), ('obj_w_lin', obj_w_lin)):
        if val is not None:
            _lib.fill(getattr(env, field), np.asarray(val).reshape(n_s))
    return env


def _host(x) -> Optional[np.ndarray]:
    return None if x is None else x.detach().cpu().numpy()


FAILURE_DUMP_PATH = 'negative_variance_state.pt'   # the reference's file name (gp_reachability_pytorch.py:266)


def save_failure_state(ssm, state: Tensor, action: Tensor, path: Optional[str] = None) -> Optional[str]:
    """What the reference's ``_save_model`` writes before it raises on a numerical failure
    (gp_reachability_pytorch.py:256-266): the GP's hyper-parameter state, the offending states and actions and the
    training data, as one ``torch.save`` file in the working directory.  Returns the path (None if the model has no
    state to save or the file cannot be written: the dump must never mask the error it accompanies)."""
    path = path or FAILURE_DUMP_PATH
    try:
        sd = ssm.state_dict() if hasattr(ssm, 'state_dict') else {}
        cpu = lambda t: None if t is None else t.detach().cpu()
        torch.save({'gp_model': sd.get('gp_model', {}), 'gp_likelihood': sd.get('gp_likelihood', {}),
                    'state': cpu(state), 'action': cpu(action), 'x_train': cpu(getattr(ssm, 'x_train', None)),
                    'y_train': cpu(getattr(ssm, 'y_train', None))}, path)
        return path
    except Exception as exc:   # noqa: BLE001 -- diagnostics only
        print(f'WARNING: could not write {path}: {exc}')
        return None


def raise_for_status(status: int, where: str, dump=None) -> None:
    """Maps the device status word onto the reference's failure modes.  `dump`, if given, is called before the
    ValueError of a numerical failure is raised (the reference saves the model state first, :78-80, :256-266)."""
    if status & _lib.SX_STATUS_NAN:
        if dump is not None:
            dump()
        raise ValueError(f'nan in {where} (sigm_0 / rkhs_bounds / b_sigma_eps): numerical failure in the GP variance')
    if status & _lib.SX_STATUS_UB_NONPOS:
        raise AssertionError('All elements of u_b must be >0')
    if status & _lib.SX_STATUS_ZERO_FIX:
        print(f'WARNING: found 0 in {where} but carried on')


def onestep_reachability(p_center: Tensor, ssm: CemSSM, k_ff: Tensor, l_mu: Tensor, l_sigma: Tensor,
                         q_shape: Optional[Tensor] = None, k_fb: Tensor = None, c_safety: float = 1., verbose: int = 1,
                         a: Tensor = None, b: Tensor = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Over-approximates the one-step reachable set under u = k_fb (x - p) + k_ff.

    p_center [N x n_s], k_ff [N x n_u], l_mu/l_sigma [n_s], q_shape [N x n_s x n_s] or None (all states are points),
    k_fb [n_u x n_s], a [n_s x n_s] / b [n_s x n_u] linear prior (default identity / zero).
    Returns (p_1 [N x n_s], q_1 [N x n_s x n_s], sigma [N x n_s]).
    """
    n = p_center.shape[0]
    n_s, n_u = ssm.num_states, ssm.num_actions
    assert_shape(p_center, (n, n_s))
    assert_shape(k_ff, (n, n_u))
    assert_shape(l_mu, (n_s,))
    assert_shape(l_sigma, (n_s,))
    assert_shape(q_shape, (n, n_s, n_s), ignore_if_none=True)
    assert_shape(k_fb, (n_u, n_s), ignore_if_none=True)
    assert_shape(a, (n_s, n_s), ignore_if_none=True)
    assert_shape(b, (n_s, n_u), ignore_if_none=True)
    _lib.require_gpu(p_center, 'p_center')
    if q_shape is not None and k_fb is None:
        raise ValueError('k_fb is required when the state is an ellipsoid')
    dev = p_center.device

    env = make_env(n_s, n_u, a=_host(a), b=_host(b) if a is not None else None, k_fb=_host(k_fb), l_mu=_host(l_mu),
                   l_sigma=_host(l_sigma), beta=c_safety)
    p = p_center.detach().contiguous()
    u = k_ff.detach().contiguous()
    if q_shape is None:
        mean, var = ssm.predict_without_jacobians(p, u)
        jac, q = None, None
    else:
        mean, var, jac = ssm.predict_with_jacobians(p, u)
        jac = jac.detach().contiguous()
        q = q_shape.detach().contiguous()
    p1 = torch.empty_like(p)
    q1 = torch.empty((n, n_s, n_s), dtype=torch.float64, device=dev)
    sigma = torch.empty_like(p)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().sx_onestep_reach(ctypes.byref(env), n, _lib.ptr(p), _lib.ptr(q), _lib.ptr(u),
                                           _lib.ptr(mean.detach().contiguous()), _lib.ptr(var.detach().contiguous()),
                                           _lib.ptr(jac), _lib.ptr(p1), _lib.ptr(q1), _lib.ptr(sigma),
                                           _lib.ptr(status), _lib.stream_ptr(dev)), 'sx_onestep_reach')
    raise_for_status(int(status.item()), 'onestep_reachability', dump=lambda: save_failure_state(ssm, p, u))
    return p1, q1, sigma


def lin_ellipsoid_safety_distance(p_center: Tensor, q_shape: Tensor, h_mat: Tensor, h_vec: Tensor,
                                  c_safety: float = 1.0) -> Tensor:
    """d [N x m] = h_mat p + c_safety sqrt(diag(h_mat Q h_mat^T)) - h_vec; d < 0 everywhere <=> certified inside."""
    return _polytope(p_center, q_shape, h_mat, h_vec, c_safety)[0]


def is_ellipsoid_inside_polytope(p_center: Tensor, q_shape: Tensor, h_mat: Tensor, h_vec: Tensor) -> Tensor:
    """bool [N]: no distance >= 0."""
    return _polytope(p_center, q_shape, h_mat, h_vec, 1.0)[1]


def _polytope(p_center, q_shape, h_mat, h_vec, c_safety):
    n = p_center.size(0)
    m, n_s = h_mat.shape
    assert_shape(p_center, (n, n_s))
    assert_shape(q_shape, (n, n_s, n_s))
    assert_shape(h_vec, (m, 1))
    _lib.require_gpu(p_center, 'p_center')
    dev = p_center.device
    env = make_env(n_s, 1, h_mat=_host(h_mat), h_vec=_host(h_vec))
    d = torch.empty((n, m), dtype=torch.float64, device=dev)
    inside = torch.empty((n,), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().sx_polytope_distance(ctypes.byref(env), n, _lib.ptr(p_center.detach().contiguous()),
                                               _lib.ptr(q_shape.detach().contiguous()), float(c_safety), _lib.ptr(d),
                                               _lib.ptr(inside), _lib.stream_ptr(dev)), 'sx_polytope_distance')
    return d, inside.bool()
