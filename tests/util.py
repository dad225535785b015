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
