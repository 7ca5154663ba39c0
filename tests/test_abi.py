"""CPU: libsxamd.so loads without a GPU, exports every symbol include/sx_amd.h declares, and the ctypes mirrors of its
structs have the C layout.  No compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sx_amd.h')


@pytest.fixture(scope='module')
def lib():
    from safe_exploration_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sx_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported(lib):
    from safe_exploration_amd import _lib
    names = declared_functions()
    assert len(names) >= 8
    for name in names:
        assert hasattr(lib, name), f'{name} is declared in include/sx_amd.h but not exported'
    assert sorted(_lib.SIGNATURES) == names, 'ctypes signatures and header out of sync'


def test_version_string(lib):
    assert lib.sx_version().decode().startswith('sxamd ') and 'gfx950' in lib.sx_version().decode()


def test_struct_layouts_match_c(tmp_path):
    """sizeof/offsetof as gcc sees the header == what ctypes computes."""
    from safe_exploration_amd import _lib
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sx_amd.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sx_gp_model), offsetof(sx_gp_model, inv_ls2),'
                   'offsetof(sx_gp_model, x_train), offsetof(sx_gp_model, stage_tab), sizeof(sx_env),'
                   'offsetof(sx_env, beta), offsetof(sx_env, obj_w_lin));'
                   'printf("%d %d %d %d %d\\n", SX_MAX_NS, SX_MAX_NU, SX_MAX_M, SX_TILE, SX_WAVES);'
                   'printf("%zu %zu %zu %zu %zu %d %d\\n", sizeof(sx_feat_model), offsetof(sx_feat_model, width),'
                   'offsetof(sx_feat_model, prelu), offsetof(sx_feat_model, noise), offsetof(sx_feat_model, minv),'
                   'SX_FEAT_MAX_WIDTH, SX_FEAT_MAX_LAYERS);'
                   'printf("%zu %zu %zu %zu %d %d\\n", sizeof(sx_mlp_model), offsetof(sx_mlp_model, predict_std),'
                   'offsetof(sx_mlp_model, width), offsetof(sx_mlp_model, masks), SX_MLP_MAX_HIDDEN, SX_MLP_MAX_WIDTH);return 0;}')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [int(v) for v in out]
    M, E = _lib.SxGpModel, _lib.SxEnv
    want = [ctypes.sizeof(M), M.inv_ls2.offset, M.x_train.offset, M.stage_tab.offset, ctypes.sizeof(E), E.beta.offset,
            E.obj_w_lin.offset, _lib.SX_MAX_NS, _lib.SX_MAX_NU, _lib.SX_MAX_M, _lib.SX_TILE, 8]
    Fm = _lib.SxFeatModel
    want += [ctypes.sizeof(Fm), Fm.width.offset, Fm.prelu.offset, Fm.noise.offset, Fm.minv.offset, _lib.SX_FEAT_MAX_WIDTH,
             _lib.SX_FEAT_MAX_LAYERS]
    Mm = _lib.SxMlpModel
    want += [ctypes.sizeof(Mm), Mm.predict_std.offset, Mm.width.offset, Mm.masks.offset, _lib.SX_MLP_MAX_HIDDEN,
             _lib.SX_MLP_MAX_WIDTH]
    assert got == want


def test_pack_sizes_is_pure_host(lib):
    a, t = ctypes.c_int64(), ctypes.c_int64()
    assert lib.sx_gp_pack_sizes(2, 1, 200, ctypes.byref(a), ctypes.byref(t)) == 0
    n_pad = (200 + 1 + 3 + 15) // 16 * 16
    nrb = n_pad // 16
    assert a.value == 2 * nrb * (nrb + 1) * 128 and t.value > 0 and t.value % 4 == 0
    assert lib.sx_gp_pack_sizes(9, 1, 200, ctypes.byref(a), ctypes.byref(t)) != 0   # n_s beyond SX_MAX_NS
    assert lib.sx_gp_pack_sizes(2, 1, 0, ctypes.byref(a), ctypes.byref(t)) != 0


def test_argument_errors_without_a_gpu(lib):
    from safe_exploration_amd import _lib
    env = _lib.SxEnv()
    # null pointers / bad sizes are rejected before anything touches the device
    assert lib.sx_cem_rank_refit(1, 4, 8, 3, None, None, 1, None, 3, None, None, None, None, None, None, None) == _lib.SX_ERR_ARG
    assert lib.sx_onestep_reach(ctypes.byref(env), 4, None, None, None, None, None, None, None, None, None, None, None) \
        == _lib.SX_ERR_ARG
    assert lib.sx_gp_predict(None, None, 1, None, None, None, None, 0, None) == _lib.SX_ERR_ARG


def test_missing_library_fails_loudly(monkeypatch):
    from safe_exploration_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libsxamd.so')
    with pytest.raises(_lib.SxError, match='no CPU fallback'):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'safe_exploration_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f


def test_rank_kernel_choice_depends_on_the_shape_only():
    """sx_cem_rank_counts (host code, no GPU): one or two problems of up to 8192 candidates are ranked by counting over the
    whole chip, anything else by one workgroup per problem -- the rule every rank of a multi-GPU solve must agree on."""
    from safe_exploration_amd import _lib
    lib = _lib.lib()
    if os.environ.get('SX_RANK_PATH'):
        pytest.skip('SX_RANK_PATH forces a kernel')
    want = {(1, 64): 1, (1, 4096): 1, (2, 4096): 1, (3, 4096): 1, (4, 4096): 0, (8, 4096): 0, (1, 8192): 1, (1, 8193): 0,
            (1, 16384): 0, (64, 16): 1, (65, 16): 0, (0, 16): 0, (1, 0): 0, (1, 3281): 1}
    for (E, P), w in want.items():
        assert int(lib.sx_cem_rank_counts(E, P)) == w, (E, P)
