"""Shared helpers for the parity tests (HIP engine vs CPU oracle on identical inputs)."""
import numpy as np

from oracle import fql_oracle as O


def make_cfg(hidden=(64, 64, 64, 64), **kw):
    import fql_amd
    cfg = fql_amd.get_config()
    cfg.update(actor_hidden_dims=tuple(hidden), value_hidden_dims=tuple(hidden), alpha=10.0)
    cfg.update(kw)
    return cfg


def randomize_params(params, seed, scale=0.1):
    """Biases / LN params away from their 0/1 init so every leaf matters; target != critic."""
    rng = np.random.default_rng(seed)
    out = O.tree_map(lambda a: np.array(a, dtype=np.float64), params)
    for path, leaf in O.tree_leaves_with_path(out):
        if path.endswith('/bias') or path.endswith('/scale'):
            leaf += scale * rng.standard_normal(leaf.shape)
    out['modules_target_critic'] = O.tree_map(lambda a: a + 0.01 * rng.standard_normal(a.shape), out['modules_critic'])
    return O.tree_map(lambda a: a.astype(np.float32), out)


def make_problem(od, ad, B, hidden, seed=0, **cfgkw):
    cfg = make_cfg(hidden, batch_size=B, **cfgkw)
    ds = O.make_synthetic_dataset(max(4 * B, 64), od, ad, seed=seed)
    idx = np.random.default_rng(seed + 1).integers(0, len(ds['observations']), size=B)
    batch = O.sample_batch(ds, idx)
    noise = O.make_noise(B, ad, seed + 2)
    return cfg, ds, batch, noise


def leaf_dict(tree):
    return dict(O.tree_leaves_with_path(tree))


def assert_info_close(got, want, rtol=2e-5, atol=2e-6, keys=None):
    for k in (keys or O.INFO_KEYS):
        g, w = float(got[k]), float(want[k])
        assert abs(g - w) <= atol + rtol * abs(w), (k, g, w)


def assert_step_matches(agent, ref, cfg, batch, noise, step=None):
    """One update on both sides: 13 infos, every leaf's gradient (Adam mu / 0.1 after the first step from zero moments), nu, post-step
    parameters.  `step(agent, batch, noise)` runs the GPU side (default: agent.update) and returns the infos or None (then read_info)."""
    _, _, g_ref = ref.grads(batch, noise)
    info_u = step(agent, batch, noise) if step is not None else agent.update(batch, noise=noise)[1]
    if info_u is None:
        info_u = agent.read_info()
    _, info_ru = ref.update(batch, noise)
    assert_info_close(info_u, info_ru, rtol=5e-5, atol=5e-6)
    opt = agent.get_opt_state()
    assert opt['count'] == 1 and opt['step'] == 2
    mu, nu = leaf_dict(opt['mu']), leaf_dict(opt['nu'])
    new, new_ref = leaf_dict(agent.get_params()), leaf_dict(ref.params)
    lr = cfg['lr']
    worst = (0.0, None)
    for p, g in leaf_dict(g_ref).items():
        scale = np.abs(g).max()
        tol = 2e-5 * scale + 1e-9
        err = np.abs(mu[p] / 0.1 - g).max()
        worst = max(worst, (err / max(scale, 1e-30), p))
        np.testing.assert_allclose(mu[p] / 0.1, g, rtol=0, atol=tol, err_msg=f'grad {p}')
        np.testing.assert_allclose(nu[p] / 0.001, g * g, rtol=1e-4, atol=tol * scale + 1e-12, err_msg=f'nu {p}')
        d = np.abs(new[p] - new_ref[p])
        stable = np.abs(g) > 50 * tol
        if 'target' in p:
            assert d.max() <= 1e-6, p
        else:
            assert d[stable].max(initial=0) <= 2e-6, (p, d[stable].max())
            assert d.max() <= 2 * lr + 1e-6, p
    return worst
