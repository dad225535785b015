"""CPU: the oracle reproduces the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import fql_oracle as O

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*.npz')) if 'visual' not in os.path.basename(p) and not os.path.basename(p).startswith('traj_'))
VISUAL_GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'visual_small.npz')
VISUAL_FULL = os.path.join(os.path.dirname(__file__), 'golden', 'visual_full.npz')      # 64x64x9, hidden 512 x 4, alpha 300, B 64


def load_case(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z['meta']))
    cfg = O.get_config()
    cfg.update(actor_hidden_dims=tuple(meta['hidden']), value_hidden_dims=tuple(meta['hidden']), batch_size=meta['B'])
    cfg.update(meta['cfg'])

    def tree(prefix):
        t = {}
        for k in z.files:
            if k.startswith(prefix + '/'):
                node = t
                keys = k[len(prefix) + 1:].split('/')
                for kk in keys[:-1]:
                    node = node.setdefault(kk, {})
                node[keys[-1]] = z[k]
        return t

    return dict(meta=meta, cfg=cfg, params=tree('params'), grads=tree('grads'), new_params=tree('new_params'),
                batch={k[6:]: z[k] for k in z.files if k.startswith('batch/')},
                noise={k[6:]: z[k] for k in z.files if k.startswith('noise/')},
                total_loss=float(z['total_loss']), info_total_loss=z['info_total_loss'], info_update=z['info_update'],
                sample_actions=z['sample_actions'], flow_actions=z['flow_actions'])


def test_golden_files_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize('path', GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize('dtype,rtol', [(np.float64, 1e-6), (np.float32, 2e-4)])
def test_oracle_reproduces_golden(path, dtype, rtol):
    c = load_case(path)
    m = c['meta']
    ref = O.OracleFQL(c['params'], c['cfg'], m['obs_dim'], m['act_dim'], dtype)
    loss, info = ref.total_loss(c['batch'], c['noise'])
    assert abs(float(loss) - c['total_loss']) <= rtol * max(1, abs(c['total_loss']))
    for i, k in enumerate(O.INFO_KEYS[:10]):
        assert abs(float(info[k]) - c['info_total_loss'][i]) <= rtol * max(1, abs(c['info_total_loss'][i])), k
    _, _, g = ref.grads(c['batch'], c['noise'])
    for (p, a), (_, b) in zip(O.tree_leaves_with_path(g), O.tree_leaves_with_path(c['grads'])):
        np.testing.assert_allclose(a, b, rtol=0, atol=rtol * max(np.abs(b).max(), 1e-6) * 5, err_msg=p)
    np.testing.assert_allclose(ref.sample_actions(c['batch']['observations'], c['noise']['eps2']), c['sample_actions'], atol=1e-5)
    np.testing.assert_allclose(ref.compute_flow_actions(c['batch']['observations'], c['noise']['z']), c['flow_actions'], atol=2e-5)
    _, iu = ref.update(c['batch'], c['noise'])
    for i, k in enumerate(O.INFO_KEYS):
        assert abs(iu[k] - c['info_update'][i]) <= rtol * max(1, abs(c['info_update'][i])), k


def grad_sample(leaves):
    """The fixture's strided sample of every gradient leaf (make_golden.visual_case): <= 64 elements per leaf, concatenated."""
    return np.concatenate([np.asarray(g).reshape(-1)[::max(1, g.size // 64)][:64].astype(np.float64) for g in leaves])


def load_visual_case(path=VISUAL_GOLDEN):
    """tests/golden/visual_*.npz: inputs + expected outputs; the parameters are regenerated from the stored seed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(os.path.dirname(__file__), 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    z = np.load(path, allow_pickle=False)
    m = json.loads(str(z['meta']))
    cfg = O.get_config()
    cfg.update(actor_hidden_dims=tuple(m['hidden']), value_hidden_dims=tuple(m['hidden']), batch_size=m['B'], alpha=m['alpha'],
               encoder='impala_small')
    params = mg.visual_params(m['seed'], m['hw'], m['c'], m['act_dim'], cfg)
    batch = {k[6:]: z[k] for k in z.files if k.startswith('batch/')}
    noise = {k[6:]: z[k] for k in z.files if k.startswith('noise/')}
    return dict(meta=m, cfg=cfg, params=params, batch=batch, noise=noise, z=z)


@pytest.mark.parametrize('dtype,rtol', [(np.float64, 1e-6), (np.float32, 5e-4)])
def test_oracle_reproduces_visual_golden(dtype, rtol):
    c = load_visual_case()
    m, z = c['meta'], c['z']
    ref = O.OracleFQL(c['params'], c['cfg'], (m['hw'], m['hw'], m['c']), m['act_dim'], dtype)
    loss, info = ref.total_loss(c['batch'], c['noise'])
    assert abs(loss - float(z['total_loss'])) <= rtol * abs(float(z['total_loss']))
    _, _, g = ref.grads(c['batch'], c['noise'])
    leaves = O.tree_leaves_with_path(g)
    assert [p for p, _ in leaves] == m['paths']
    l2 = np.array([np.sqrt(np.sum(np.square(v.astype(np.float64)))) for _, v in leaves])
    np.testing.assert_allclose(l2, z['grad_l2'], rtol=20 * rtol, atol=1e-9)
    np.testing.assert_allclose(ref.sample_actions(c['batch']['observations'], c['noise']['eps2']), z['sample_actions'], atol=2e-4 if dtype == np.float32 else 1e-6)
    np.testing.assert_allclose(ref.compute_flow_actions(c['batch']['observations'], c['noise']['z']), z['flow_actions'], atol=2e-4 if dtype == np.float32 else 1e-6)
    _, iu = ref.update(c['batch'], c['noise'])
    for i, k in enumerate(O.INFO_KEYS):
        assert abs(iu[k] - z['info_update'][i]) <= 10 * rtol * abs(z['info_update'][i]) + 1e-6, k


def test_oracle_reproduces_full_size_visual_golden():
    """tests/golden/visual_full.npz (BASELINE configs[4]'s shapes: uint8 [64,64,64,9], impala_small, hidden 512 x 4, alpha 300): the
    fp32 oracle against the committed fp64 values -- loss terms, per-leaf gradient norms and the strided element sample."""
    c = load_visual_case(VISUAL_FULL)
    m, z = c['meta'], c['z']
    assert (m['hw'], m['c'], m['B'], m['hidden']) == (64, 9, 64, [512] * 4)
    ref = O.OracleFQL(c['params'], c['cfg'], (m['hw'], m['hw'], m['c']), m['act_dim'], np.float32)
    _, info, g = ref.grads(c['batch'], c['noise'])
    for i, k in enumerate(O.INFO_KEYS[:10]):
        assert abs(info[k] - z['info_total_loss'][i]) <= 5e-4 * max(1.0, abs(z['info_total_loss'][i])), k
    leaves = O.tree_leaves_with_path(g)
    assert [p for p, _ in leaves] == m['paths']
    l2 = np.array([np.sqrt(np.sum(np.square(v.astype(np.float64)))) for _, v in leaves])
    np.testing.assert_allclose(l2, z['grad_l2'], rtol=1e-2, atol=1e-9)
    gs = grad_sample([v for _, v in leaves])
    scale = np.repeat(z['grad_max'], [min(64, len(v.reshape(-1)[::max(1, v.size // 64)])) for _, v in leaves])
    assert np.mean(np.abs(gs - z['grad_sample']) <= 2e-3 * scale + 1e-12) >= 0.99
