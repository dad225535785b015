"""GPU parity: the HIP engine (through the C ABI) vs the CPU oracle on identical params/batch/noise.

Tolerances (fp32 path, "vs CPU restatement of the reference", parity unpinned at the JAX boundary):
  * info scalars: |d| <= 2e-6 + 2e-5*|ref|   (north_star asks losses within 1e-4)
  * gradients (read back as Adam mu/0.1 after one step): <= 2e-5 * max|g_leaf| + 1e-9 per element
  * post-step params: 1e-6 where the oracle gradient is well away from 0; Adam's first step is
    -lr*sign(g) so elements with |g| ~ 0 may differ by up to 2*lr.
"""
import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import assert_info_close, leaf_dict, make_problem, randomize_params

pytestmark = pytest.mark.gpu


def _agent(cfg, batch, params=None, seed=0):
    import fql_amd
    agent = fql_amd.FQLAgent.create(seed, batch['observations'][:1], batch['actions'][:1], cfg)
    if params is not None:
        agent.set_params(params)
    return agent


CASES = [
    # od, ad, B, hidden, cfg overrides
    (7, 3, 16, (32, 32, 32, 32), {}),
    (29, 8, 64, (64, 64, 64, 64), {}),
    (29, 8, 32, (48, 80, 64), {'q_agg': 'min'}),              # ragged widths, 3 hidden layers, padding
    (11, 5, 32, (64, 64, 64, 64), {'normalize_q_loss': True}),
    (40, 4, 48, (64, 64), {'actor_layer_norm': True, 'flow_steps': 3}),
    (17, 6, 16, (32, 32, 32, 32), {'layer_norm': False}),
]


@pytest.mark.parametrize('od,ad,B,hidden,kw', CASES)
def test_total_loss_and_update_match_oracle(od, ad, B, hidden, kw):
    cfg, ds, batch, noise = make_problem(od, ad, B, hidden, seed=3, **kw)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=5)
    agent.set_params(params)
    got_back = leaf_dict(agent.get_params())
    for p, a in leaf_dict(params).items():
        np.testing.assert_array_equal(got_back[p], a, err_msg=p)   # set/get round trip is exact

    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    loss_ref, info_ref = ref.total_loss(batch, noise)
    loss, info = agent.total_loss(batch, noise=noise)
    assert abs(loss - float(loss_ref)) <= 2e-6 + 2e-5 * abs(float(loss_ref))
    assert_info_close(info, info_ref, keys=O.INFO_KEYS[:10])
    # total_loss must not change any state
    for p, a in leaf_dict(agent.get_params()).items():
        np.testing.assert_array_equal(a, leaf_dict(params)[p], err_msg=p)

    _, _, g_ref = ref.grads(batch, noise)
    _, info_u = agent.update(batch, noise=noise)
    _, info_ru = ref.update(batch, noise)
    assert_info_close(info_u, info_ru, rtol=5e-5, atol=5e-6)
    opt = agent.get_opt_state()
    assert opt['count'] == 1 and opt['step'] == 2
    mu = leaf_dict(opt['mu']); nu = leaf_dict(opt['nu'])
    new = leaf_dict(agent.get_params()); new_ref = leaf_dict(ref.params)
    lr = cfg['lr']
    for p, g in leaf_dict(g_ref).items():
        scale = np.abs(g).max()
        tol = 2e-5 * scale + 1e-9
        np.testing.assert_allclose(mu[p] / 0.1, g, rtol=0, atol=tol, err_msg=f'grad {p}')
        np.testing.assert_allclose(nu[p] / 0.001, g * g, rtol=1e-4, atol=tol * scale + 1e-12, err_msg=f'nu {p}')
        d = np.abs(new[p] - new_ref[p])
        stable = np.abs(g) > 50 * tol
        if 'target' in p:
            assert d.max() <= 1e-6, p
        else:
            assert d[stable].max(initial=0) <= 2e-6, (p, d[stable].max())
            assert d.max() <= 2 * lr + 1e-6, p


def test_polyak_reads_pre_step_critic():
    cfg, ds, batch, noise = make_problem(7, 3, 16, (32, 32, 32, 32), seed=11)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=2)
    agent.set_params(params)
    agent.update(batch, noise=noise)
    new = agent.get_params()
    for p, a in leaf_dict(params['modules_critic']).items():
        t_old = leaf_dict(params['modules_target_critic'])[p]
        np.testing.assert_allclose(leaf_dict(new['modules_target_critic'])[p], 0.005 * a + 0.995 * t_old, rtol=3e-7, atol=1e-7, err_msg=p)


def test_multi_step_trajectory_tracks_oracle():
    od, ad, B = 29, 8, 64
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=21)
    agent = _agent(cfg, batch)
    params = agent.get_params()
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float32)
    rng = np.random.default_rng(0)
    for s in range(20):
        idx = rng.integers(0, len(ds['observations']), size=B)
        b = O.sample_batch(ds, idx)
        nz = O.make_noise(B, ad, 100 + s)
        _, ig = agent.update(b, noise=nz)
        _, ir = ref.update(b, nz)
        assert_info_close(ig, ir, rtol=2e-3, atol=2e-4)
    opt = agent.get_opt_state()
    assert opt['count'] == 20 and opt['step'] == 21


def test_full_size_config_one_update():
    """BASELINE.json configs[1]: obs=29, act=8, B=256, hidden 512x4, flow_steps=10, alpha=10."""
    od, ad, B = 29, 8, 256
    cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=31)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=7, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    loss_ref, info_ref = ref.total_loss(batch, noise)
    loss, info = agent.total_loss(batch, noise=noise)
    assert_info_close(info, info_ref, keys=O.INFO_KEYS[:10])
    _, info_u = agent.update(batch, noise=noise)
    _, info_ru = ref.update(batch, noise)
    assert_info_close(info_u, info_ru, rtol=5e-5, atol=5e-6)
    st = agent.stats()
    assert st['macs_per_update'] == 24171520 * 256 and st['param_count'] == 4871700


def test_sample_and_flow_actions():
    od, ad, B = 29, 8, 64
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=41)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=9)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    for n in (1, 5, 16, 37, 64):
        obs = batch['observations'][:n]
        z = noise['z'][:n]
        np.testing.assert_allclose(agent.sample_actions(obs, noises=z), ref.sample_actions(obs, z), atol=2e-6)
        np.testing.assert_allclose(agent.compute_flow_actions(obs, z), ref.compute_flow_actions(obs, z), atol=5e-6)
    one = agent.sample_actions(batch['observations'][0], noises=noise['z'][0])   # 1-D obs, main.py:225
    assert one.shape == (ad,)
    np.testing.assert_allclose(one, ref.sample_actions(batch['observations'][0], noise['z'][0]), atol=2e-6)
    # engine RNG path: deterministic per seed, clipped, different across seeds
    a1 = agent.sample_actions(batch['observations'], seed=np.array([1, 2], dtype=np.uint32))
    a2 = agent.sample_actions(batch['observations'], seed=np.array([1, 2], dtype=np.uint32))
    a3 = agent.sample_actions(batch['observations'], seed=np.array([1, 3], dtype=np.uint32))
    np.testing.assert_array_equal(a1, a2)
    assert np.abs(a1).max() <= 1 and np.abs(a1 - a3).max() > 1e-3


def test_errors_are_python_exceptions():
    cfg, ds, batch, noise = make_problem(7, 3, 16, (32, 32), seed=1)
    agent = _agent(cfg, batch)
    bad = dict(batch); bad['observations'] = batch['observations'][:, :5]
    with pytest.raises(ValueError):
        agent.update(bad)
    with pytest.raises(KeyError):
        agent.set_params({'modules_nope': {'x': np.zeros(3, np.float32)}})
    with pytest.raises(ValueError):
        agent.set_params({'modules_critic': {'value_net': {'Dense_0': {'bias': np.zeros(3, np.float32)}}}})
    with pytest.raises(ValueError):
        agent.sample_actions(np.zeros((4, 9), np.float32))


def test_config3_cube_shaped_batch_1024():
    """BASELINE.json configs[2]: cube-single-shaped (obs=40, act=4), alpha=300, batch=1024."""
    od, ad, B = 40, 4, 1024
    cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=51, alpha=300.0)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=8, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float32)   # fp32 numpy oracle: ~10 s at this size
    _, info_u = agent.update(batch, noise=noise)
    _, info_ru = ref.update(batch, noise)
    assert_info_close(info_u, info_ru, rtol=2e-4, atol=2e-5)
    assert agent.stats()['macs_per_update'] == 24227840 * 1024     # 48,455,680 FLOP/sample


def test_batch_size_switch_keeps_state():
    od, ad = 11, 5
    cfg, ds, batch, noise = make_problem(od, ad, 64, (64, 64, 64, 64), seed=61)
    agent = _agent(cfg, batch)
    before = leaf_dict(agent.get_params())
    small = {k: v[:32] for k, v in batch.items()}
    loss, _ = agent.total_loss(small, noise={k: v[:32] for k, v in noise.items()})   # re-sizes the workspace to 32 rows
    assert np.isfinite(loss)
    for p, a in leaf_dict(agent.get_params()).items():
        np.testing.assert_array_equal(a, before[p], err_msg=p)
    agent.update(batch, noise=noise)                                                  # and back to 64
    assert agent.get_opt_state()['count'] == 1


def test_config3_shape_b1024_matches_oracle():
    """BASELINE configs[2] shape (obs 40, act 4, alpha 300, B 1024, hidden 512x4): at this size the engine's defaults switch
    to 64x64 side tiles; two updates against the fp64 oracle."""
    import fql_amd
    od, ad, B = 40, 4, 1024
    cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=31, alpha=300.0)
    agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    params = randomize_params(agent.get_params(), seed=6, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    for s in range(2):
        nz = O.make_noise(B, ad, 70 + s)
        _, ig = agent.update(batch, noise=nz)
        _, ir = ref.update(batch, nz)
        assert_info_close(ig, ir, rtol=1e-4, atol=1e-5)
