"""ctypes binding of libsxamd.so (include/sx_amd.h).  There is no CPU fallback: a missing library is an error."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_int, c_int32, c_int64, c_uint8, c_void_p

import numpy as np
import torch

SX_MAX_NS = 4
SX_MAX_NU = 2
SX_MAX_D = SX_MAX_NS + SX_MAX_NU
SX_MAX_M = 16
SX_TILE = 16
SX_FEAT_MAX_WIDTH = 32
SX_FEAT_MAX_LAYERS = 3
SX_MLP_MAX_HIDDEN = 4
SX_MLP_MAX_WIDTH = 64

SX_OK, SX_ERR_ARG, SX_ERR_UNSUPPORTED, SX_ERR_LAUNCH = 0, 1, 2, 3
SX_STATUS_NAN, SX_STATUS_ZERO_FIX, SX_STATUS_UB_NONPOS, SX_STATUS_NOT_PD = 1, 2, 4, 8
SX_OBJ_NEG_VARIANCE, SX_OBJ_AFFINE_ABS = 0, 1
SX_CON_TERMINAL, SX_CON_ALL_STATES = 0, 1
SX_PROF_ROLLOUT_FUSED, SX_PROF_RANK, SX_PROF_KSTAR_BIG, SX_PROF_TRMM_BIG, SX_PROF_STEP_BIG, SX_PROF_ROLLOUT_FEAT, SX_PROF_ROLLOUT_MLP = range(7)
PROF_KERNELS = {SX_PROF_ROLLOUT_FUSED: 'cem_rollout_kernel', SX_PROF_RANK: 'cem_rank_kernel',
                SX_PROF_KSTAR_BIG: 'kstar_big_kernel', SX_PROF_TRMM_BIG: 'trmm_reduce_kernel',
                SX_PROF_STEP_BIG: 'step_big_kernel', SX_PROF_ROLLOUT_FEAT: 'cem_rollout_feat_kernel',
                SX_PROF_ROLLOUT_MLP: 'cem_rollout_mlp_kernel'}
SX_ACTION_VIOLATION_COST, SX_STATE_VIOLATION_COST = 3.0, 10.0

_ERR = {SX_ERR_ARG: 'bad argument (null pointer, non-positive size or inconsistent shapes)',
        SX_ERR_UNSUPPORTED: 'unsupported dimension (n_s/n_u not instantiated, too many polytope rows, or the training '
                            'set does not fit the fused kernel\'s LDS budget)',
        SX_ERR_LAUNCH: 'HIP launch error'}


class SxGpModel(Structure):
    _fields_ = [('n_s', c_int32), ('n_u', c_int32), ('n_train', c_int32), ('n_pad', c_int32),
                ('inv_ls2', c_double * (SX_MAX_NS * SX_MAX_D)), ('outputscale', c_double * SX_MAX_NS),
                ('noise', c_double * SX_MAX_NS), ('x_train', c_void_p), ('a_pack', c_void_p), ('stage_tab', c_void_p)]


class SxFeatModel(Structure):
    _fields_ = [('n_s', c_int32), ('n_u', c_int32), ('n_feat', c_int32), ('n_layers', c_int32), ('normalise', c_int32),
                ('width', c_int32 * (SX_FEAT_MAX_LAYERS + 1)), ('prelu', c_double), ('noise', c_double * SX_MAX_NS),
                ('net', c_void_p), ('wbar', c_void_p), ('minv', c_void_p)]


class SxMlpModel(Structure):
    _fields_ = [('n_s', c_int32), ('n_u', c_int32), ('n_hidden', c_int32), ('n_out', c_int32), ('n_samples', c_int32),
                ('predict_std', c_int32), ('width', c_int32 * (SX_MLP_MAX_HIDDEN + 1)), ('net', c_void_p),
                ('masks', c_void_p)]


class SxEnv(Structure):
    _fields_ = [('n_s', c_int32), ('n_u', c_int32), ('m', c_int32), ('obj_mode', c_int32), ('con_mode', c_int32),
                ('reserved', c_int32), ('beta', c_double),
                ('a', c_double * (SX_MAX_NS * SX_MAX_NS)), ('b', c_double * (SX_MAX_NS * SX_MAX_NU)),
                ('k_fb', c_double * (SX_MAX_NU * SX_MAX_NS)), ('l_mu', c_double * SX_MAX_NS),
                ('l_sigma', c_double * SX_MAX_NS), ('h_mat', c_double * (SX_MAX_M * SX_MAX_NS)),
                ('h_vec', c_double * SX_MAX_M), ('u_min', c_double * SX_MAX_NU), ('u_max', c_double * SX_MAX_NU),
                ('obj_w_abs', c_double * SX_MAX_NS), ('obj_target', c_double * SX_MAX_NS),
                ('obj_w_lin', c_double * SX_MAX_NS)]


# name -> (restype, argtypes): exactly the entry points include/sx_amd.h declares
SIGNATURES = {
    'sx_version': (c_char_p, []),
    'sx_gp_pack_sizes': (c_int, [c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    'sx_gp_fit': (c_int, [POINTER(SxGpModel)] + [c_void_p] * 7),
    'sx_gp_mll_grad': (c_int, [POINTER(SxGpModel)] + [c_void_p] * 8),
    'sx_gp_pack': (c_int, [POINTER(SxGpModel), c_void_p, c_void_p, c_void_p]),
    'sx_gp_predict': (c_int, [POINTER(SxGpModel), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                              c_void_p]),
    'sx_gp_predict_workspace_bytes': (c_int64, [POINTER(SxGpModel), c_int]),
    'sx_gp_predict_var_jac': (c_int, [POINTER(SxGpModel), c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'sx_gp_predict_mean_hessian': (c_int, [POINTER(SxGpModel), c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'sx_feat_features': (c_int, [POINTER(SxFeatModel), c_void_p, c_int, c_void_p, c_void_p]),
    'sx_feat_fit': (c_int, [POINTER(SxFeatModel), c_void_p, c_void_p, c_int, POINTER(c_double)] + [c_void_p] * 5),
    'sx_feat_predict': (c_int, [POINTER(SxFeatModel), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'sx_cem_rollout_feat': (c_int, [POINTER(SxFeatModel), POINTER(SxEnv), c_int, c_int, c_int] + [c_void_p] * 12),
    'sx_mlp_predict': (c_int, [POINTER(SxMlpModel), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'sx_cem_rollout_mlp': (c_int, [POINTER(SxMlpModel), POINTER(SxEnv), c_int, c_int, c_int] + [c_void_p] * 12),
    'sx_onestep_reach': (c_int, [POINTER(SxEnv), c_int] + [c_void_p] * 11),
    'sx_polytope_distance': (c_int, [POINTER(SxEnv), c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p,
                                     c_void_p]),
    'sx_cem_rollout': (c_int, [POINTER(SxGpModel), POINTER(SxEnv), c_int, c_int, c_int] + [c_void_p] * 12
                       + [c_int64, c_void_p]),
    'sx_cem_rollout_elites': (c_int, [POINTER(SxGpModel), POINTER(SxEnv), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_int] + [c_void_p] * 10),
    'sx_cem_rollout_workspace_bytes': (c_int64, [POINTER(SxGpModel), c_int, c_int, c_int]),
    'sx_profile_enable': (c_int, [c_int]),
    'sx_profile_stride': (c_int, [c_int]),
    'sx_profile_stride_kind': (c_int, [c_int, c_int]),
    'sx_profile_collect': (c_int, [c_int, POINTER(c_double), POINTER(c_int64)]),
    'sx_profile_disable': (c_int, []),
    'sx_cem_rank_counts': (c_int, [c_int, c_int]),
    'sx_cem_rollout_form': (c_int, [POINTER(SxGpModel), c_int]),
    'sx_cem_pack_result': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'sx_cem_rank_refit': (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64]
                          + [c_void_p] * 7),
}

# SX_LIB selects a diagnostic build of the same library (tools/phase_stamps.py); the default is the product build
LIB_PATH = os.environ.get('SX_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libsxamd.so')
_lib = None


class SxError(RuntimeError):
    pass


def lib():
    """Loads libsxamd.so once.  Raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SxError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          f'(safe_exploration_amd has no CPU fallback)')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != SX_OK:
        raise SxError(f'{what}: {_ERR.get(code, code)}')


def ptr(t):
    """Device pointer of a contiguous float64/int tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'libsxamd takes contiguous row-major buffers'
    return c_void_p(t.data_ptr())


def stream_ptr(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t, name: str):
    if not t.is_cuda:
        raise SxError(f'{name} must live on the GPU: safe_exploration_amd has no CPU path (got device {t.device})')
    if t.dtype != torch.float64:
        raise SxError(f'{name} must be float64 (the reference runs in double precision), got {t.dtype}')


def fill(carray, values) -> None:
    flat = np.asarray(values, dtype=np.float64).reshape(-1)
    assert len(flat) <= len(carray), (len(flat), len(carray))
    for i, v in enumerate(flat):
        carray[i] = float(v)


def profile_collect() -> dict:
    """{kernel name: (total ms, launches)} of the launches recorded since sx_profile_enable (synchronises them)."""
    out = {}
    for kind, name in PROF_KERNELS.items():
        ms, n = c_double(), c_int64()
        check(lib().sx_profile_collect(kind, ctypes.byref(ms), ctypes.byref(n)), 'sx_profile_collect')
        if n.value:
            out[name] = (ms.value, n.value)
    return out
