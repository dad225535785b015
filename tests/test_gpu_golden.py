"""GPU: the HIP engine (through the C ABI) reproduces the committed golden vectors."""
import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.test_golden_oracle import GOLDEN, VISUAL_FULL, grad_sample, load_case, load_visual_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('path', GOLDEN)
def test_engine_reproduces_golden(path):
    import fql_amd
    c = load_case(path)
    m = c['meta']
    cfg = fql_amd.get_config()
    cfg.update({k: v for k, v in c['cfg'].items() if k in cfg})
    agent = fql_amd.FQLAgent.create(0, c['batch']['observations'][:1], c['batch']['actions'][:1], cfg)
    agent.set_params(c['params'])
    loss, info = agent.total_loss(c['batch'], noise=c['noise'])
    assert abs(loss - c['total_loss']) <= 2e-6 + 2e-5 * abs(c['total_loss'])
    for i, k in enumerate(O.INFO_KEYS[:10]):
        assert abs(info[k] - c['info_total_loss'][i]) <= 2e-6 + 2e-5 * abs(c['info_total_loss'][i]), k
    np.testing.assert_allclose(agent.sample_actions(c['batch']['observations'], noises=c['noise']['eps2']), c['sample_actions'], atol=2e-6)
    np.testing.assert_allclose(agent.compute_flow_actions(c['batch']['observations'], c['noise']['z']), c['flow_actions'], atol=5e-6)
    _, iu = agent.update(c['batch'], noise=c['noise'])
    for i, k in enumerate(O.INFO_KEYS):
        assert abs(iu[k] - c['info_update'][i]) <= 5e-6 + 5e-5 * abs(c['info_update'][i]), k
    mu = dict(O.tree_leaves_with_path(agent.get_opt_state()['mu']))
    for p, g in O.tree_leaves_with_path(c['grads']):
        np.testing.assert_allclose(mu[p] / 0.1, g, rtol=0, atol=2e-5 * np.abs(g).max() + 1e-9, err_msg=p)


def test_engine_reproduces_visual_golden():
    """tests/golden/visual_small.npz (impala_small encoders on uint8 images): info scalars, per-leaf gradient norms, actions."""
    import fql_amd
    c = load_visual_case()
    m, z = c['meta'], c['z']
    agent = fql_amd.FQLAgent.create(0, c['batch']['observations'][:1], c['batch']['actions'][:1], c['cfg'])
    agent.set_params(c['params'])
    loss, info = agent.total_loss(c['batch'], noise=c['noise'])
    assert abs(loss - float(z['total_loss'])) <= 2e-5 + 1e-4 * abs(float(z['total_loss']))
    for i, k in enumerate(O.INFO_KEYS[:10]):
        assert abs(info[k] - z['info_total_loss'][i]) <= 2e-5 + 1e-4 * abs(z['info_total_loss'][i]), k
    np.testing.assert_allclose(agent.sample_actions(c['batch']['observations'], noises=c['noise']['eps2']), z['sample_actions'], atol=5e-6)
    np.testing.assert_allclose(agent.compute_flow_actions(c['batch']['observations'], c['noise']['z']), z['flow_actions'], atol=2e-5)
    _, iu = agent.update(c['batch'], noise=c['noise'])
    for i, k in enumerate(O.INFO_KEYS):
        assert abs(iu[k] - z['info_update'][i]) <= 2e-5 + 1e-4 * abs(z['info_update'][i]), k
    mu = dict(O.tree_leaves_with_path(agent.get_opt_state()['mu']))
    assert list(mu) == m['paths']
    l2 = np.array([np.sqrt(np.sum(np.square(v.astype(np.float64) / 0.1))) for v in mu.values()])
    # per-leaf gradient norms; loose enough for a max-pool tie-break, tight for everything else
    np.testing.assert_allclose(l2, z['grad_l2'], rtol=2e-2, atol=1e-9)
    assert np.mean(np.abs(l2 - z['grad_l2']) <= 1e-4 * z['grad_l2'] + 1e-9) >= 0.75


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_engine_reproduces_full_size_visual_golden(precision):
    """tests/golden/visual_full.npz: BASELINE configs[4]'s shapes (uint8 [64,64,64,9], impala_small, hidden 512 x 4, alpha 300) against
    committed fp64 oracle values: 13 infos, both action functions, per-leaf gradient norms and a strided sample of every leaf's gradient.
    The allowance for the norms and the sample is test_gpu_visual.py's: a max-pool window whose two largest inputs differ by less than
    fp32 rounding may route its gradient to the other input, which moves single elements of the conv gradients
    upstream of that pool and nothing else."""
    import fql_amd
    c = load_visual_case(VISUAL_FULL)
    m, z = c['meta'], c['z']
    cfg = dict(c['cfg']); cfg['precision'] = precision
    agent = fql_amd.FQLAgent.create(0, c['batch']['observations'][:1], c['batch']['actions'][:1], cfg)
    agent.set_params(c['params'])
    rel = 1e-4 if precision == 'fp32' else 5e-4
    loss, info = agent.total_loss(c['batch'], noise=c['noise'])
    assert abs(loss - float(z['total_loss'])) <= rel * abs(float(z['total_loss']))
    for i, k in enumerate(O.INFO_KEYS[:10]):
        assert abs(info[k] - z['info_total_loss'][i]) <= rel * max(1.0, abs(z['info_total_loss'][i])), k
    atol = 2e-5 if precision == 'fp32' else 2e-4
    np.testing.assert_allclose(agent.sample_actions(c['batch']['observations'], noises=c['noise']['eps2']), z['sample_actions'], atol=atol)
    np.testing.assert_allclose(agent.compute_flow_actions(c['batch']['observations'], c['noise']['z']), z['flow_actions'], atol=atol)
    _, iu = agent.update(c['batch'], noise=c['noise'])
    for i, k in enumerate(O.INFO_KEYS):
        assert abs(iu[k] - z['info_update'][i]) <= rel * max(1.0, abs(z['info_update'][i])), k
    mu = dict(O.tree_leaves_with_path(agent.get_opt_state()['mu']))
    assert list(mu) == m['paths']
    grads = [v.astype(np.float64) / 0.1 for v in mu.values()]                     # zero moments before the step: mu = 0.1 x gradient
    l2 = np.array([np.sqrt(np.sum(np.square(g))) for g in grads])
    np.testing.assert_allclose(l2, z['grad_l2'], rtol=2e-2, atol=1e-9)
    assert np.mean(np.abs(l2 - z['grad_l2']) <= (1e-4 if precision == 'fp32' else 1e-3) * z['grad_l2'] + 1e-9) >= 0.75
    gs = grad_sample(grads)
    scale = np.repeat(z['grad_max'], [len(g.reshape(-1)[::max(1, g.size // 64)][:64]) for g in grads])
    err = np.abs(gs - z['grad_sample'])
    ok = err <= (1e-4 if precision == 'fp32' else 1e-3) * scale + 1e-12
    dense = np.repeat([('stack_blocks' not in p) for p in m['paths']], [len(g.reshape(-1)[::max(1, g.size // 64)][:64]) for g in grads])
    assert ok[dense].all()                  # no pool or ReLU-after-conv upstream of these leaves: every sampled element
    # conv leaves: fp32 measured 1 pool tie in 63 leaves (every other element within 1e-6 of the leaf's largest); bf16x3 rounds the
    # activations differently, so more ReLU signs / pool winners near zero flip: medians <= 1.7e-3, worst element 1.3e-2 (experiments/golden_err.py)
    assert (err[~dense] <= 3e-2 * scale[~dense] + 1e-12).all()
    if precision == 'fp32':
        assert ok.mean() >= 0.98, (ok.mean(), err.max())
