"""GPU: checkpoint round trip in the reference's state-dict layout (utils/flax_utils.py:162-202) and the opt-in
persistent Euler-chain kernel against the oracle."""
import os

import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import assert_info_close, leaf_dict, make_problem, randomize_params

pytestmark = pytest.mark.gpu


def test_checkpoint_roundtrip_resumes_bit_identically(tmp_path):
    import fql_amd
    from fql_amd import checkpoint
    od, ad, B = 29, 8, 32
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=13)
    a = fql_amd.FQLAgent.create(3, batch['observations'][:1], batch['actions'][:1], cfg)
    for s in range(3):
        a.update(batch, noise=O.make_noise(B, ad, 50 + s))
    sd = checkpoint.to_state_dict(a)
    # layout of flax.serialization.to_state_dict(FQLAgent)
    assert set(sd) == {'rng', 'network'} and set(sd['network']) == {'step', 'params', 'opt_state'}
    assert set(sd['network']['opt_state']) == {'0', '1'} and set(sd['network']['opt_state']['0']) == {'count', 'mu', 'nu'}
    assert int(sd['network']['step']) == 4 and int(sd['network']['opt_state']['0']['count']) == 3
    assert sd['network']['params']['modules_critic']['value_net']['Dense_0']['kernel'].shape == (2, od + ad, 64)
    checkpoint.save_agent(a, str(tmp_path), 7)
    b = fql_amd.FQLAgent.create(99, batch['observations'][:1], batch['actions'][:1], cfg)
    b = checkpoint.restore_agent(b, str(tmp_path), 7)
    for p, x in leaf_dict(a.get_params()).items():
        np.testing.assert_array_equal(leaf_dict(b.get_params())[p], x, err_msg=p)
    nz = O.make_noise(B, ad, 77)
    _, ia = a.update(batch, noise=nz)
    _, ib = b.update(batch, noise=nz)
    assert ia == ib                                     # same state, same kernels: bitwise equal infos
    assert b.get_opt_state()['count'] == 4


@pytest.mark.parametrize('B', [256, 512])
def test_persistent_euler_chain_matches_oracle(B):
    """FQL_PEC=1: the whole Euler chain in one persistent launch (team hand-offs as {tag, value} granules);
    B=512 gives every team two row tiles, which it runs through each phase in turn."""
    import fql_amd
    od, ad = 29, 8
    os.environ['FQL_PEC'] = '1'
    try:
        cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=17)
        agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    finally:
        del os.environ['FQL_PEC']
    params = randomize_params(agent.get_params(), seed=4, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    for s in range(3):
        nz = O.make_noise(B, ad, 90 + s)
        _, ig = agent.update(batch, noise=nz)
        _, ir = ref.update(batch, nz)
        assert_info_close(ig, ir, rtol=1e-4, atol=1e-5)
    assert agent.stats()['launches_per_update'] < 60     # the chain is one launch on this path (30 in the default program)


def test_jax_key_paths_match_oracle_with_host_threefry_noise():
    """config['rng']='jax': update() draws its noise with the reference's key derivation; sample_actions(seed=key)
    with a JAX key draws the reference's noise.  Checked against the oracle fed the same host-generated tensors."""
    import fql_amd
    from fql_amd import jax_prng as J
    od, ad, B = 29, 8, 32
    cfg, ds, batch, _ = make_problem(od, ad, B, (64, 64, 64, 64), seed=29)
    cfg['rng'] = 'jax'
    agent = fql_amd.FQLAgent.create(5, batch['observations'][:1], batch['actions'][:1], cfg)
    np.testing.assert_array_equal(agent.rng, J.split(J.PRNGKey(5), 2)[0])           # agents/fql.py:189-190
    ref = O.OracleFQL(agent.get_params(), {k: v for k, v in cfg.items() if k != 'rng'}, od, ad, np.float64)
    rng = agent.rng
    for _ in range(2):
        rng, nz = J.fql_update_noise(rng, B, ad)
        _, ig = agent.update(batch)
        _, ir = ref.update(batch, nz)
        assert_info_close(ig, ir, rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(agent.rng, rng)
    key = np.array([123, 456], dtype=np.uint32)
    got = agent.sample_actions(batch['observations'], seed=key)
    want = ref.sample_actions(batch['observations'], J.sample_actions_noise(key, (B,), ad))
    np.testing.assert_allclose(got, want, atol=1e-5)


@pytest.mark.parametrize('mode', ['jax', 'jax_partitionable'])
def test_jax_noise_generated_on_the_device_matches_the_host_restatement(mode):
    """fql_noise_from_jax_keys: the five tensors of an update from the reference's five keys, both threefry layouts.  Bits are exact
    (t, a plain mantissa fill, must be EQUAL); normals go through the single-precision erf_inv polynomial on the device and scipy's
    float64 erfinv on the host: <= 6e-7 * (1 + |x|).  The update with device noise equals the update given the host tensors."""
    import fql_amd
    from fql_amd import jax_prng as J
    from tests.test_gpu_hardening import _draws
    part = mode == 'jax_partitionable'
    od, ad, B = 11, 5, 48                                  # (batch sizes are multiples of 16: the odd-size padding of the original layout cannot occur)
    cfg, ds, batch, _ = make_problem(od, ad, B, (64, 64, 64), seed=61)
    cfg['rng'] = mode
    a = fql_amd.FQLAgent.create(9, batch['observations'][:1], batch['actions'][:1], cfg)
    cfg_h = fql_amd.get_config(); cfg_h.update(dict(cfg)); cfg_h['rng_device'] = False
    b = fql_amd.FQLAgent.create(9, batch['observations'][:1], batch['actions'][:1], cfg_h)
    np.testing.assert_array_equal(a.rng, J.split(J.PRNGKey(9), 2, partitionable=part)[0])
    rng = a.rng
    for _ in range(2):
        rng, keys = J.fql_update_keys(rng, partitionable=part)
        want = J.noise_from_keys(keys, B, ad, partitionable=part)
        _, ia = a.update(batch)
        got, _, _ = _draws(a, B, od, ad)
        np.testing.assert_array_equal(got['t'], want['t'])
        for k in ('eps1', 'z', 'eps2'):
            assert np.abs(got[k] - want[k]).max() <= 6e-7 * (1 + np.abs(want[k]).max()), k
        assert np.abs(got['x0'] - want['x0']).max() <= 2e-6          # recovered as a - (a - x0): one more rounding
        _, ib = b.update(batch)
        assert_info_close(ia, ib, rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(a.rng, rng)
    np.testing.assert_array_equal(b.rng, rng)
