"""JAX-compatible PRNG on the host (numpy): threefry2x32, split, uniform, normal.

Purpose: let reference call sites keep passing JAX keys (`agent.sample_actions(obs, seed=key)`,
`agent.rng`) and obtain the SAME noise tensors the reference would draw, so sampled actions / losses can be compared
seed-for-seed with the JAX reference once real JAX outputs are available.

This follows JAX's public definitions for the default, non-partitionable threefry implementation
(`jax_threefry_partitionable=False`, the default over most of the version range the reference allows,
requirements.txt:2): Threefry-2x32 with 20 rounds (Salmon et al. 2011; rotation constants 13,15,26,6 / 17,29,16,24,
key-schedule constant 0x1BD11BDA), `split` = threefry over iota(2n) reshaped to (n, 2), `random_bits` = threefry over
iota(size), `uniform` = mantissa bits | 1.0f minus 1, `normal` = sqrt(2) * erfinv(uniform(-1+eps, 1)).

PARITY UNPINNED: jax is not installable here, so this cannot be run against JAX.  It is pinned only by known-answer
values quoted in JAX's public documentation (tests/test_jax_prng.py); XLA's float32 erf_inv polynomial is replaced by
scipy's erfinv (differences ~1e-7 relative).  Results under `jax_threefry_partitionable=True` differ by design.
"""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M32 = np.uint64(0xFFFFFFFF)


def PRNGKey(seed: int) -> np.ndarray:
    """jax.random.PRNGKey: [high 32 bits, low 32 bits] of the (64-bit) seed."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([seed >> 32, seed & 0xFFFFFFFF], dtype=np.uint32)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def threefry_2x32(key, count) -> np.ndarray:
    """Threefry-2x32 of a flat uint32 `count` array under `key` (uint32[2]); returns uint32 of the same length."""
    key = np.asarray(key, dtype=np.uint32).reshape(2)
    count = np.asarray(count, dtype=np.uint32).reshape(-1)
    n = count.size
    if n % 2:
        count = np.concatenate([count, np.zeros(1, np.uint32)])
    half = count.size // 2
    x0 = count[:half].astype(np.uint64)
    x1 = count[half:].astype(np.uint64)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    ks = (k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & _M32)
    x0 = (x0 + ks[0]) & _M32
    x1 = (x1 + ks[1]) & _M32
    for i in range(5):
        for r in _ROT[i % 2]:
            x0 = (x0 + x1) & _M32
            x1 = _rotl(x1, r)
            x1 ^= x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M32
        x1 = (x1 + ks[(i + 2) % 3] + np.uint64(i + 1)) & _M32
    return np.concatenate([x0, x1]).astype(np.uint32)[:n]


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split: uint32[num, 2]."""
    return threefry_2x32(key, np.arange(2 * num, dtype=np.uint32)).reshape(num, 2)


def random_bits(key, shape) -> np.ndarray:
    size = int(np.prod(shape)) if len(shape) else 1
    return threefry_2x32(key, np.arange(size, dtype=np.uint32)).reshape(shape)


def uniform(key, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    """jax.random.uniform(float32)."""
    bits = random_bits(key, tuple(shape))
    fl = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    minval, maxval = np.float32(minval), np.float32(maxval)
    return np.maximum(minval, fl * (maxval - minval) + minval).astype(np.float32)


def normal(key, shape=()) -> np.ndarray:
    """jax.random.normal(float32): sqrt(2) * erfinv(u), u ~ U(nextafter(-1, 0), 1)."""
    from scipy.special import erfinv
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape, minval=lo, maxval=1.0)
    return (np.float32(np.sqrt(2.0)) * erfinv(u.astype(np.float64))).astype(np.float32)


def fql_update_noise(rng, batch_size: int, action_dim: int):
    """The five noise tensors FQLAgent.update(batch) draws from agent.rng, with the reference's key derivation
    (agents/fql.py:125 -> :100 -> :24,143-150 / :49-54,62-63,82).  Returns (new_rng, noise dict)."""
    new_rng, rng = split(rng)                               # update():        new_rng, rng = split(self.rng)
    return new_rng, fql_total_loss_noise(rng, batch_size, action_dim)


def fql_total_loss_noise(rng, batch_size: int, action_dim: int):
    """Noise of FQLAgent.total_loss(batch, grad_params, rng) (agents/fql.py:94-111): rng itself is the argument (or
    agent.rng when None, as the validation probe main.py:284 does)."""
    _, actor_rng, critic_rng = split(rng, 3)                # total_loss():    rng, actor_rng, critic_rng = split(rng, 3)
    _, sample_rng = split(critic_rng)                       # critic_loss():   rng, sample_rng = split(rng)
    eps1 = normal(split(sample_rng)[0], (batch_size, action_dim))      # sample_actions(): action_seed, _ = split(seed)
    r, x_rng, t_rng = split(actor_rng, 3)                   # actor_loss():    rng, x_rng, t_rng = split(rng, 3)
    x0 = normal(x_rng, (batch_size, action_dim))
    t = uniform(t_rng, (batch_size, 1))
    r, noise_rng = split(r)                                 #                  rng, noise_rng = split(rng)
    z = normal(noise_rng, (batch_size, action_dim))
    eps2 = normal(split(r)[0], (batch_size, action_dim))    # sample_actions(batch['observations'], seed=rng)
    return dict(eps1=eps1, x0=x0, t=t.reshape(-1), z=z, eps2=eps2)


def sample_actions_noise(seed, lead_shape, action_dim: int) -> np.ndarray:
    """The noise FQLAgent.sample_actions(obs, seed) draws: normal(split(seed)[0], (*obs.shape[:-1], action_dim))."""
    return normal(split(seed)[0], tuple(lead_shape) + (action_dim,))
