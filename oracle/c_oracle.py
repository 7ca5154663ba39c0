"""ctypes wrapper of the C restatement (oracle/csrc/oracle_rollout.c).  TEST INFRASTRUCTURE - see oracle/__init__.py.

Same inputs and outputs as ``oracle.cem.rollout``; OpenMP over particles.  It exists (a) as a second, independently
written check of the rollout arithmetic and (b) as a CPU baseline that is not dominated by Python overhead.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_double, c_int, c_void_p

import numpy as np

from .cem import Problem, RolloutResult

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
_PATH = os.path.join(_DIR, 'liboracle.so')
_lib = None


class _Problem(Structure):
    _fields_ = [('n_s', c_int), ('n_u', c_int), ('n', c_int), ('m', c_int), ('obj_mode', c_int), ('con_mode', c_int),
                ('beta', c_double)] + [(name, c_void_p) for name in (
                    'x', 'chol', 'alpha', 'ls', 'os', 'noise', 'a', 'b', 'kfb', 'l_mu', 'l_sigma', 'h_mat', 'h_vec',
                    'u_min', 'u_max', 'w_abs', 'target', 'w_lin')]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(['make', '-s', '-C', _DIR])
        handle = ctypes.CDLL(_PATH)
        handle.sxo_rollout.restype = c_int
        handle.sxo_rollout.argtypes = [POINTER(_Problem), c_int, c_int] + [c_void_p] * 8
        handle.sxo_max_threads.restype = c_int
        _lib = handle
    return _lib


def max_threads() -> int:
    return int(lib().sxo_max_threads())


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rollout(prob: Problem, gp, x0, actions, q0=None, want_traj=True) -> RolloutResult:
    """gp: oracle.gp.ExactGP.  x0 [n_s]; actions [P x H x n_u]."""
    P, H, n_u = actions.shape
    n_s = prob.n_s
    zeros = np.zeros(n_s)
    keep = dict(x=_c(gp.X), chol=_c(np.stack(gp.L)), alpha=_c(np.stack(gp.alpha)), ls=_c(gp.ls), os=_c(gp.s),
                noise=_c(gp.noise), a=_c(prob.a), b=_c(prob.b), kfb=_c(prob.k_fb), l_mu=_c(prob.l_mu),
                l_sigma=_c(prob.l_sigma), h_mat=_c(prob.h_mat), h_vec=_c(prob.h_vec).reshape(-1), u_min=_c(prob.u_min),
                u_max=_c(prob.u_max), w_abs=_c(prob.obj_w_abs if prob.obj_w_abs is not None else zeros),
                target=_c(prob.obj_target if prob.obj_target is not None else zeros),
                w_lin=_c(prob.obj_w_lin if prob.obj_w_lin is not None else zeros))
    cp = _Problem(n_s, n_u, gp.n, prob.h_mat.shape[0], prob.obj_mode, prob.con_mode, float(prob.beta),
                  **{k: v.ctypes.data for k, v in keep.items()})
    acts = _c(actions)
    x0c = _c(x0)
    q0c = None if q0 is None else _c(q0)
    out = RolloutResult(np.empty((P, H, n_s)) if want_traj else None, np.empty((P, H, n_s, n_s)) if want_traj else None,
                        np.empty((P, H, n_s)) if want_traj else None, np.empty(P), np.empty(P))
    ptr = lambda arr: None if arr is None else arr.ctypes.data
    st = lib().sxo_rollout(ctypes.byref(cp), P, H, x0c.ctypes.data, ptr(q0c), acts.ctypes.data, ptr(out.traj_p),
                           ptr(out.traj_q), ptr(out.sigma), out.obj_cost.ctypes.data, out.con_cost.ctypes.data)
    if st < 0:
        raise ValueError('dimension beyond the C oracle\'s limits')
    out.status = st
    return out
