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




def visual_params(seed, hw, c, ad, cfg):
    """Deterministic parameters of the visual golden case (numpy's PCG64 streams are stable across versions): the full tree
    would be ~5 MB per copy, so the fixture stores the seed and per-leaf summaries instead of the tensors."""
    params = O.init_params(seed, (hw, hw, c), ad, cfg, np.float64)
    rng = np.random.default_rng(seed + 1)
    for path, leaf in O.tree_leaves_with_path(params):
        if path.endswith('/bias') or path.endswith('/scale'):
            leaf += 0.05 * rng.standard_normal(leaf.shape)
    params['modules_target_critic'] = O.tree_map(lambda a: a + 0.01 * rng.standard_normal(a.shape), params['modules_critic'])
    return O.tree_map(lambda a: a.astype(np.float32), params)


def visual_case(name='visual_small', hw=32, c=3, ad=4, B=16, hidden=(32, 32, 32, 32), seed=4242, alpha=3.0):
    """BASELINE configs[4] in miniature: uint8 [B,32,32,3] observations, impala_small encoders; and (visual_full) at
    its real image size and widths: [B,64,64,9], hidden 512 x 4, alpha 300, batch 64 (the smallest the engine's encoder Dense takes)."""
    cfg = O.get_config()
    cfg.update(actor_hidden_dims=hidden, value_hidden_dims=hidden, batch_size=B, alpha=alpha, encoder='impala_small')
    params = visual_params(seed, hw, c, ad, cfg)
    rng = np.random.default_rng(seed + 2)
    batch = {'observations': rng.integers(0, 256, size=(B, hw, hw, c), dtype=np.uint8),
             'next_observations': rng.integers(0, 256, size=(B, hw, hw, c), dtype=np.uint8),
             'actions': rng.uniform(-1, 1, size=(B, ad)).astype(np.float32),
             'rewards': -(rng.random(B) < 0.9).astype(np.float32), 'masks': (rng.random(B) < 0.9).astype(np.float32)}
    noise = O.make_noise(B, ad, seed + 3)
    ref = O.OracleFQL(params, cfg, (hw, hw, c), ad, np.float64)
    loss, info_tl = ref.total_loss(batch, noise)
    _, _, grads = ref.grads(batch, noise)
    sample = ref.sample_actions(batch['observations'], noise['eps2'])
    flow = ref.compute_flow_actions(batch['observations'], noise['z'])
    _, info_up = ref.update(batch, noise)
    paths = [p for p, _ in O.tree_leaves_with_path(grads)]
    out = {'meta': json.dumps(dict(hw=hw, c=c, act_dim=ad, B=B, hidden=list(hidden), seed=seed, alpha=alpha, paths=paths))}
    for k, v in batch.items():
        out[f'batch/{k}'] = v
    for k, v in noise.items():
        out[f'noise/{k}'] = v
    out['grad_l2'] = np.array([np.sqrt(np.sum(np.square(g))) for _, g in O.tree_leaves_with_path(grads)])
    out['grad_sum'] = np.array([np.sum(g) for _, g in O.tree_leaves_with_path(grads)])
    # a strided sample of every leaf's gradient (<= 64 elements each): element-level evidence without 20 MB of tensors
    out['grad_sample'] = np.concatenate([g.reshape(-1)[::max(1, g.size // 64)][:64].astype(np.float64) for _, g in O.tree_leaves_with_path(grads)])
    out['grad_max'] = np.array([np.abs(g).max() for _, g in O.tree_leaves_with_path(grads)])
    out['new_param_sum'] = np.array([np.sum(v.astype(np.float64)) for _, v in O.tree_leaves_with_path(ref.params)])
    out['total_loss'] = np.float64(loss)
    out['info_total_loss'] = np.array([float(info_tl[k]) for k in O.INFO_KEYS[:10]])
    out['info_update'] = np.array([float(info_up[k]) for k in O.INFO_KEYS])
    out['sample_actions'] = sample.astype(np.float32)
    out['flow_actions'] = flow.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(name, 'loss', float(loss), 'bytes', os.path.getsize(os.path.join(HERE, f'{name}.npz')))


if __name__ == '__main__':
    if '--visual-full-only' in sys.argv:
        visual_case('visual_full', hw=64, c=9, ad=5, B=64, hidden=(512, 512, 512, 512), seed=4343, alpha=300.0)
        sys.exit(0)
    if '--visual-only' not in sys.argv:
        main()
    visual_case()
    visual_case('visual_full', hw=64, c=9, ad=5, B=64, hidden=(512, 512, 512, 512), seed=4343, alpha=300.0)
