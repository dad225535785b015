"""CPU: the C-ABI library loads and exports every symbol include/fql_amd.h declares; the host mirror
validates arguments; nothing here launches GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from fql_amd import _cabi
    return _cabi.load()


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'fql_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(fql_[a-z_0-9]+)\s*\(', src)))


def test_header_symbols_are_exported_and_bound(lib):
    from fql_amd import _cabi
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/fql_amd.h but not exported by libfql_amd.so'
        assert n in _cabi.SYMBOLS, f'{n} declared in the header but not bound in fql_amd/_cabi.py'
    for n in _cabi.SYMBOLS:
        assert n in names, f'{n} bound in _cabi.py but not declared in the header'


def test_abi_version_config_defaults_and_info_names(lib):
    from fql_amd import _cabi, INFO_KEYS
    assert lib.fql_abi_version() == 1
    c = _cabi.FqlConfig()
    lib.fql_default_config(C.byref(c))
    # agents/fql.py:249-270 defaults
    assert (c.num_actor_hidden, c.num_value_hidden) == (4, 4)
    assert list(c.actor_hidden)[:4] == [512] * 4 and list(c.value_hidden)[:4] == [512] * 4
    assert c.layer_norm == 1 and c.actor_layer_norm == 0
    assert abs(c.lr - 3e-4) < 1e-10 and abs(c.discount - 0.99) < 1e-7 and abs(c.tau - 0.005) < 1e-9
    assert c.alpha == 300.0 and c.q_agg == 0 and c.flow_steps == 10 and c.normalize_q_loss == 0 and c.batch_size == 256
    assert C.sizeof(_cabi.FqlConfig) == 4 * (2 + 1 + 8 + 1 + 8 + 2 + 4 + 4 + 1 + 7)
    assert tuple(lib.fql_info_name(i).decode() for i in range(13)) == INFO_KEYS


def test_python_config_matches_reference_defaults():
    import fql_amd
    cfg = fql_amd.get_config()
    assert cfg['lr'] == 3e-4 and cfg['batch_size'] == 256 and cfg['alpha'] == 300.0 and cfg['tau'] == 0.005
    assert cfg['actor_hidden_dims'] == (512,) * 4 and cfg['q_agg'] == 'mean' and cfg['flow_steps'] == 10
    assert cfg['layer_norm'] is True and cfg['actor_layer_norm'] is False and cfg['encoder'] is None
    assert fql_amd.agents['fql'] is fql_amd.FQLAgent
    assert cfg.lr == cfg['lr']


def test_create_without_gpu_fails_loudly_or_succeeds_on_gpu(lib):
    """No CPU fallback: on a box without a HIP device fql_create must fail with FQL_E_NODEVICE."""
    import torch
    from fql_amd import _cabi
    c = _cabi.FqlConfig()
    lib.fql_default_config(C.byref(c))
    c.obs_dim, c.act_dim, c.batch_size = 5, 2, 16
    for i in range(4):
        c.actor_hidden[i] = c.value_hidden[i] = 32
    h = C.c_void_p()
    rc = lib.fql_create(C.byref(c), 0, C.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        lib.fql_destroy(h)
    else:
        assert rc == _cabi.FQL_E_NODEVICE
        assert b'no CPU fallback' in lib.fql_last_error(None)
        import fql_amd
        with pytest.raises(RuntimeError):
            fql_amd.FQLAgent.create(0, np.zeros((1, 5), np.float32), np.zeros((1, 2), np.float32), dict(batch_size=16))


def test_null_handles_are_rejected(lib):
    assert lib.fql_num_leaves(None) < 0
    assert lib.fql_set_batch_size(None, 16) < 0
    assert lib.fql_read_info(None, None) < 0


def test_mirror_rejects_bad_config_before_touching_the_device():
    import fql_amd
    ob, ac = np.zeros((1, 5), np.float32), np.zeros((1, 2), np.float32)
    with pytest.raises(ValueError):                     # an encoder needs image observations [H, W, C]
        fql_amd.FQLAgent.create(0, ob, ac, dict(encoder='impala_small'))
    with pytest.raises(NotImplementedError):            # only impala_small (BASELINE config 5) is built
        fql_amd.FQLAgent.create(0, np.zeros((1, 64, 64, 9), np.uint8), ac, dict(encoder='impala_large'))
    with pytest.raises(ValueError):
        fql_amd.FQLAgent.create(0, np.zeros((1, 8, 8, 3), np.float32), ac, {})
    with pytest.raises(ValueError):
        fql_amd.FQLAgent.create(0, ob, ac, dict(q_agg='median'))


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under fql_amd/ may import, call or link it."""
    import re
    pkg = os.path.join(ROOT, 'fql_amd')
    pat = re.compile(r'^\s*(from\s+\.*oracle|import\s+oracle|from\s+\S*fql_oracle|import\s+\S*fql_oracle)|oracle/|fql_oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), (dirpath, f)
