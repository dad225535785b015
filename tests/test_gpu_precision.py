"""precision = 'bf16x3' (fql_config.precision = 2; SURVEY 8b config key `precision`, 8d "bf16x3-split"): the dense contractions of the
side lanes (32-row tiles, weight gradients) and of the Euler chain run as split-bf16 products on the bf16 matrix cores.

What is asserted, against the fp64 CPU restatement of the reference on identical batches / noise / parameters:
* the 13 infos within 1e-4 max(1, |ref|) - BASELINE.json's "losses within 1e-4" - at BASELINE configs[1] and configs[2] shapes, at
  hidden 256 (the other chain-kernel instantiation) and at small ragged shapes (32 x 32 tiles, K = 64 single-chunk tiles);
  measured (experiments/precision_probe.py): <= 2e-5 absolute on every loss, <= 1.2e-4 absolute on grad/norm ~ 84;
* every leaf's gradient within 1e-4 max|g_leaf| (measured worst 1.2e-5 at configs[1], 2.9e-5 at hidden 64; the fp32 mode: 6e-7);
* a 20-update trajectory stays within 5e-4 of the oracle's (the fp32 mode's bound in test_gpu_parity.py is 2e-3 ... same test shape);
* the mode is a property of the handle: precision 1 (plain bf16) is refused.
Not JAX: "vs CPU restatement of the reference" (parity unpinned at the JAX boundary).
"""
import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import leaf_dict, make_problem, randomize_params

pytestmark = pytest.mark.gpu


def _agent(cfg, batch, seed=0):
    import fql_amd
    return fql_amd.FQLAgent.create(seed, batch['observations'][:1], batch['actions'][:1], cfg)


def _one_update(od, ad, B, hidden, alpha, precision, **kw):
    cfg, ds, batch, noise = make_problem(od, ad, B, hidden, seed=41, alpha=alpha, precision=precision, **kw)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=9, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    _, _, g_ref = ref.grads(batch, noise)
    _, info = agent.update(batch, noise=noise)
    _, info_r = ref.update(batch, noise)
    mu = leaf_dict(agent.get_opt_state()['mu'])
    worst = (0.0, None)
    for p, g in leaf_dict(g_ref).items():
        sc = np.abs(g).max()
        worst = max(worst, (float(np.abs(mu[p] / 0.1 - g).max() / max(sc, 1e-30)), p))
    deltas = {k: abs(float(info[k]) - float(info_r[k])) / max(1.0, abs(float(info_r[k]))) for k in O.INFO_KEYS}
    return worst, deltas


CASES = [
    (29, 8, 256, (512, 512, 512, 512), 10.0, {}),                       # BASELINE configs[1]
    (40, 4, 1024, (512, 512, 512, 512), 300.0, {}),                     # BASELINE configs[2] shape
    (29, 8, 64, (256, 256, 256, 256), 10.0, {}),                        # fql_chain_split_kernel<256>
    (29, 8, 64, (64, 64, 64, 64), 10.0, {}),                            # single-chunk tiles, generic Euler path
    (17, 6, 32, (80, 48, 64, 32), 10.0, {}),                            # ragged widths
    (29, 8, 64, (128, 128, 128, 128), 10.0, dict(q_agg='min', actor_layer_norm=True, normalize_q_loss=True)),
]


@pytest.mark.parametrize('od,ad,B,hidden,alpha,kw', CASES, ids=['configs1', 'configs2', 'h256', 'h64', 'ragged', 'ln_min_norm'])
def test_bf16x3_update_within_the_loss_bound_of_the_oracle(od, ad, B, hidden, alpha, kw):
    worst, deltas = _one_update(od, ad, B, hidden, alpha, 'bf16x3', **kw)
    assert max(deltas.values()) <= 1e-4, deltas          # north_star: losses within 1e-4 (relative to max(1, |ref|))
    assert worst[0] <= 1e-4, worst                        # max over leaves of max|g_gpu - g_ref| / max|g_ref|


def test_bf16x3_is_close_to_but_not_the_fp32_mode():
    """The two modes run different kernels: same inputs, gradients agree to ~1e-5 and are not bit-identical."""
    w32, d32 = _one_update(29, 8, 256, (512, 512, 512, 512), 10.0, 'fp32')
    wx3, dx3 = _one_update(29, 8, 256, (512, 512, 512, 512), 10.0, 'bf16x3')
    assert w32[0] <= 2e-5 and wx3[0] <= 1e-4
    assert wx3[0] > w32[0]            # the split products are less exact than fp32 fma chains; if equal, the mode was not taken


def test_bf16x3_trajectory_follows_the_oracle():
    od, ad, B = 29, 8, 64
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=5, precision='bf16x3')
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=3, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    rng = np.random.default_rng(11)
    for s in range(20):
        b = O.sample_batch(ds, rng.integers(0, len(ds['observations']), size=B))
        n = O.make_noise(B, ad, 100 + s)
        _, info = agent.update(b, noise=n)
        _, want = ref.update(b, n)
        for k in O.INFO_KEYS:
            assert abs(float(info[k]) - float(want[k])) <= 5e-4 * max(1.0, abs(float(want[k]))), (s, k, float(info[k]), float(want[k]))


def test_unknown_precisions_are_refused():
    import fql_amd
    cfg, ds, batch, noise = make_problem(29, 8, 32, (64, 64, 64, 64), seed=1)
    cfg['precision'] = 'bf16'
    with pytest.raises(ValueError):
        _agent(cfg, batch)
    from fql_amd import _cabi
    import ctypes as C
    lib = _cabi.load()
    c = _cabi.FqlConfig()
    lib.fql_default_config(C.byref(c))
    c.obs_dim, c.act_dim, c.precision = 29, 8, 1
    h = C.c_void_p()
    assert lib.fql_create(C.byref(c), 0, C.byref(h)) != 0


def test_bf16x3_visual_agent_follows_the_oracle():
    """impala_small encoders + precision 'bf16x3': the float convolutions (forward, data gradient), the encoder's Dense and the MLPs run
    split; the uint8 first convolution and the convolution weight gradients stay on the fp32 matrix cores.  Per-leaf gradients with the
    visual test's structure (tests/test_gpu_visual.py: convolutions below a max-pool may deviate through a tie-break; every other leaf is
    tight - here 2e-4 of the leaf's scale instead of 5e-5), then infos over further updates."""
    from tests.test_gpu_visual import make_visual
    from tests.util import assert_info_close
    cfg, batch, _ = make_visual(precision='bf16x3')
    B, ad = 32, 4
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=3, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), (32, 32, 3), ad, np.float64)
    nz = O.make_noise(B, ad, 50)
    _, _, g_ref = ref.grads(batch, nz)
    _, ig = agent.update(batch, noise=nz)
    _, ir = ref.update(batch, nz)
    assert_info_close(ig, ir, rtol=5e-4, atol=5e-5)
    mu = leaf_dict(agent.get_opt_state()['mu'])
    tight, loose = 0, []
    for p, g in leaf_dict(g_ref).items():
        if p.startswith('modules_target_critic'):
            continue
        scale = np.abs(g).max()
        err = np.abs(mu[p] / 0.1 - g).max()
        assert err <= 3e-2 * scale + 1e-9, (p, err / scale)
        if err <= 2e-4 * scale + 1e-9:
            tight += 1
        else:
            assert '/encoder/stack_blocks_' in p, (p, err / scale)
            loose.append(p)
    assert tight >= 3 * len(loose), loose
    for s in range(2):
        nz = O.make_noise(B, ad, 51 + s)
        _, ig = agent.update(batch, noise=nz)
        _, ir = ref.update(batch, nz)
        assert_info_close(ig, ir, rtol=1e-3, atol=1e-4)
