"""Synthetic offline data of the benchmark workloads (BASELINE.json configs[1], [2], [4]).

The reference trains on OGBench / D4RL downloads (envs/env_utils.py:103-110), which do not exist offline; the
throughput benchmark and the examples use data of the same SHAPES drawn from fixed numpy streams instead.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def make_synthetic_dataset(n: int, obs_dim: int, act_dim: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """antmaze-shaped transitions: obs ~ N(0,1), actions ~ U(-1,1) clipped as envs/env_utils.py:138-146,
    rewards -1 w.p. .99 else 0, masks 1 w.p. .99 else 0, next_obs = obs + 0.1 N(0,1), terminals = 1 - masks."""
    rng = np.random.default_rng(seed)
    obs = rng.standard_normal((n, obs_dim)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    act = np.clip(act, -1 + 1e-5, 1 - 1e-5).astype(np.float32)
    rew = np.where(rng.uniform(size=n) < 0.99, -1.0, 0.0).astype(np.float32)
    masks = np.where(rng.uniform(size=n) < 0.99, 1.0, 0.0).astype(np.float32)
    nobs = (obs + 0.1 * rng.standard_normal((n, obs_dim))).astype(np.float32)
    return dict(observations=obs, actions=act, rewards=rew, masks=masks,
                next_observations=nobs, terminals=(1.0 - masks).astype(np.float32))


def make_synthetic_frames(n: int, act_dim: int, h: int = 64, w: int = 64, c: int = 3, seed: int = 0,
                          episode_len: int = 200) -> Dict[str, np.ndarray]:
    """visual-cube-shaped replay: uint8 frames [n, h, w, c] (frame stacking happens in the gather), episodes of
    ~episode_len steps marked in `terminals` (utils/datasets.py:58-62 derives the episode starts from them)."""
    rng = np.random.default_rng(seed)
    term = (rng.random(n) < 1.0 / episode_len).astype(np.float32)
    term[-1] = 1
    return {'observations': rng.integers(0, 256, size=(n, h, w, c), dtype=np.uint8),
            'next_observations': rng.integers(0, 256, size=(n, h, w, c), dtype=np.uint8),
            'actions': np.clip(rng.uniform(-1, 1, size=(n, act_dim)), -1 + 1e-5, 1 - 1e-5).astype(np.float32),
            'rewards': -(rng.random(n) < 0.99).astype(np.float32), 'masks': 1 - term, 'terminals': term}
