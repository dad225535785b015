#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (float64).  Run from the repo root:

    python tests/golden/make_golden.py

The reference (zhouzypaul/fql) ships no fixtures and cannot be executed here (jax/flax absent), so these
vectors pin the ORACLE ("CPU restatement of the reference"), not JAX: parity stays "unpinned" at the JAX
boundary.  Each file holds inputs (params, batch, noise, config) and expected outputs (13 info
scalars of update(), the 10 of total_loss(), per-leaf gradients, post-step params, sample/flow actions).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fql_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (obs_dim, act_dim, B, hidden, cfg overrides)
    'small_mean': (7, 3, 16, (32, 32, 32, 32), dict(alpha=10.0)),
    'small_min_norm': (11, 5, 32, (48, 32, 64), dict(alpha=3.0, q_agg='min', normalize_q_loss=True)),
    'antmaze_h64': (29, 8, 64, (64, 64, 64, 64), dict(alpha=10.0)),
}


def main():
    for name, (od, ad, B, hidden, kw) in CASES.items():
        cfg = O.get_config()
        cfg.update(actor_hidden_dims=hidden, value_hidden_dims=hidden, batch_size=B)
        cfg.update(kw)
        seed = abs(hash(name)) % 1000 if False else sum(map(ord, name))
        params = O.init_params(seed, od, ad, cfg, np.float64)
        rng = np.random.default_rng(seed + 1)
        for path, leaf in O.tree_leaves_with_path(params):
            if path.endswith('/bias') or path.endswith('/scale'):
                leaf += 0.1 * rng.standard_normal(leaf.shape)
        params['modules_target_critic'] = O.tree_map(lambda a: a + 0.01 * rng.standard_normal(a.shape), params['modules_critic'])
        params = O.tree_map(lambda a: a.astype(np.float32), params)
        ds = O.make_synthetic_dataset(4 * B, od, ad, seed=seed)
        batch = O.sample_batch(ds, rng.integers(0, 4 * B, size=B))
        noise = O.make_noise(B, ad, seed + 2)
        ref = O.OracleFQL(params, cfg, od, ad, np.float64)
        loss, info_tl = ref.total_loss(batch, noise)
        _, _, grads = ref.grads(batch, noise)
        sample = ref.sample_actions(batch['observations'], noise['eps2'])
        flow = ref.compute_flow_actions(batch['observations'], noise['z'])
        _, info_up = ref.update(batch, noise)
        out = {'meta': json.dumps(dict(obs_dim=od, act_dim=ad, B=B, hidden=list(hidden), cfg={k: v for k, v in cfg.items() if k in kw or k in ('alpha', 'q_agg', 'normalize_q_loss')}))}
        for k, v in batch.items():
            out[f'batch/{k}'] = v
        for k, v in noise.items():
            out[f'noise/{k}'] = v
        for p, v in O.tree_leaves_with_path(params):
            out[f'params/{p}'] = v
        for p, v in O.tree_leaves_with_path(grads):
            out[f'grads/{p}'] = v.astype(np.float32)
        for p, v in O.tree_leaves_with_path(ref.params):
            out[f'new_params/{p}'] = v.astype(np.float32)
        out['total_loss'] = np.float64(loss)
        out['info_total_loss'] = np.array([float(info_tl[k]) for k in O.INFO_KEYS[:10]])
        out['info_update'] = np.array([float(info_up[k]) for k in O.INFO_KEYS])
        out['sample_actions'] = sample.astype(np.float32)
        out['flow_actions'] = flow.astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
        print(name, 'loss', float(loss), 'bytes', os.path.getsize(os.path.join(HERE, f'{name}.npz')))


if __name__ == '__main__':
    main()
