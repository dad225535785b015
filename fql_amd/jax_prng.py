"""JAX-compatible PRNG on the host (numpy): threefry2x32, split, uniform, normal.

Purpose: let reference call sites keep passing JAX keys (`agent.sample_actions(obs, seed=key)`,
`agent.rng`) and obtain the SAME noise tensors the reference would draw, so sampled actions / losses can be compared
seed-for-seed with the JAX reference once real JAX outputs are available.

This follows JAX's public definitions for the default, non-partitionable threefry implementation
(`jax_threefry_partitionable=False`, the default over most of the version range the reference allows,
requirements.txt:2): Threefry-2x32 with 20 rounds (Salmon et al. 2011; rotation constants 13,15,26,6 / 17,29,16,24,
key-schedule constant 0x1BD11BDA), `split` = threefry over iota(2n) reshaped to (n, 2), `random_bits` = threefry over
iota(size), `uniform` = mantissa bits | 1.0f minus 1, `normal` = sqrt(2) * erfinv(uniform(-1+eps, 1)).

Both layouts JAX has shipped are restated, selected per call (`partitionable=`) or by the module default
(`set_threefry_partitionable`, mirroring the `jax_threefry_partitionable` config flag, whose default changed from False to True
inside the version range requirements.txt:2 allows):
  * original:       split = threefry over iota(2n) reshaped (n, 2); random_bits = threefry over iota(size), the counter array cut
                    in two halves (first half -> word 0, second half -> word 1 of a block; odd sizes padded with one zero counter);
  * partitionable:  element i of any shape uses the 64-bit counter i as (hi, lo) = (i >> 32, i & 0xffffffff);
                    split(key, n)[i] = both output words of that block, random_bits[i] = word 0 XOR word 1.

PARITY UNPINNED: jax is not installable here, so this cannot be run against JAX.  The original layout is pinned by known-answer
values quoted in JAX's public documentation and the Threefry-2x32 core by the Random123 known-answer vector
(tests/test_jax_prng.py); the partitionable layout has no quoted vector available here and rests on the restatement alone.  XLA's
float32 erf_inv polynomial is replaced by scipy's erfinv on the host (differences ~1e-7 relative); the device generator
(fql_noise_from_jax_keys) evaluates the single-precision Giles polynomial XLA uses.
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


_PARTITIONABLE = False


def set_threefry_partitionable(flag: bool) -> None:
    """Module default of the layout (jax.config.update('jax_threefry_partitionable', flag))."""
    global _PARTITIONABLE
    _PARTITIONABLE = bool(flag)


def _part(partitionable):
    return _PARTITIONABLE if partitionable is None else bool(partitionable)


def threefry_block(key, x0, x1):
    """Threefry-2x32, 20 rounds, on arrays of counter words (x0, x1) under `key` (uint32[2]); returns the two output word arrays."""
    key = np.asarray(key, dtype=np.uint32).reshape(2)
    x0 = np.asarray(x0, dtype=np.uint32).astype(np.uint64)
    x1 = np.asarray(x1, dtype=np.uint32).astype(np.uint64)
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
    return x0.astype(np.uint32), x1.astype(np.uint32)


def threefry_2x32(key, count) -> np.ndarray:
    """Threefry-2x32 of a flat uint32 `count` array under `key` (uint32[2]); returns uint32 of the same length (original layout:
    the array is cut in two halves that form the two words of each block)."""
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


def split(key, num: int = 2, partitionable=None) -> np.ndarray:
    """jax.random.split: uint32[num, 2]."""
    if _part(partitionable):
        b0, b1 = threefry_block(key, np.zeros(num, np.uint32), np.arange(num, dtype=np.uint32))
        return np.stack([b0, b1], axis=1)
    return threefry_2x32(key, np.arange(2 * num, dtype=np.uint32)).reshape(num, 2)


def random_bits(key, shape, partitionable=None) -> np.ndarray:
    size = int(np.prod(shape)) if len(shape) else 1
    if _part(partitionable):
        i = np.arange(size, dtype=np.uint64)
        b0, b1 = threefry_block(key, (i >> np.uint64(32)).astype(np.uint32), (i & _M32).astype(np.uint32))
        return (b0 ^ b1).reshape(shape)
    return threefry_2x32(key, np.arange(size, dtype=np.uint32)).reshape(shape)


def uniform(key, shape=(), minval=0.0, maxval=1.0, partitionable=None) -> np.ndarray:
    """jax.random.uniform(float32)."""
    bits = random_bits(key, tuple(shape), partitionable)
    fl = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    minval, maxval = np.float32(minval), np.float32(maxval)
    return np.maximum(minval, fl * (maxval - minval) + minval).astype(np.float32)


def normal(key, shape=(), partitionable=None) -> np.ndarray:
    """jax.random.normal(float32): sqrt(2) * erfinv(u), u ~ U(nextafter(-1, 0), 1)."""
    from scipy.special import erfinv
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape, minval=lo, maxval=1.0, partitionable=partitionable)
    return (np.float32(np.sqrt(2.0)) * erfinv(u.astype(np.float64))).astype(np.float32)


def fql_update_keys(rng, partitionable=None):
    """Key derivation of FQLAgent.update(batch) (agents/fql.py:125 -> :100 -> :24,143-150 / :49-54,62-63,82) without drawing
    anything: (new_rng, {'eps1','x0','t','z','eps2': uint32[2]}) - what the device generator (fql_noise_from_jax_keys) consumes."""
    new_rng, rng = split(rng, 2, partitionable)             # update():        new_rng, rng = split(self.rng)
    return new_rng, fql_total_loss_keys(rng, partitionable)


def fql_total_loss_keys(rng, partitionable=None):
    p = partitionable
    _, actor_rng, critic_rng = split(rng, 3, p)             # total_loss():    rng, actor_rng, critic_rng = split(rng, 3)
    _, sample_rng = split(critic_rng, 2, p)                 # critic_loss():   rng, sample_rng = split(rng)
    k_eps1 = split(sample_rng, 2, p)[0]                     # sample_actions(): action_seed, _ = split(seed)
    r, x_rng, t_rng = split(actor_rng, 3, p)                # actor_loss():    rng, x_rng, t_rng = split(rng, 3)
    r, noise_rng = split(r, 2, p)                           #                  rng, noise_rng = split(rng)
    k_eps2 = split(r, 2, p)[0]                              # sample_actions(batch['observations'], seed=rng)
    return dict(eps1=k_eps1, x0=x_rng, t=t_rng, z=noise_rng, eps2=k_eps2)


def noise_from_keys(keys, batch_size: int, action_dim: int, partitionable=None):
    """The five tensors those keys draw (host): normal [B, ad] x 4 and uniform [B, 1] for t."""
    sh = (batch_size, action_dim)
    return dict(eps1=normal(keys['eps1'], sh, partitionable), x0=normal(keys['x0'], sh, partitionable),
                t=uniform(keys['t'], (batch_size, 1), partitionable=partitionable).reshape(-1), z=normal(keys['z'], sh, partitionable),
                eps2=normal(keys['eps2'], sh, partitionable))


def fql_update_noise(rng, batch_size: int, action_dim: int, partitionable=None):
    """The five noise tensors FQLAgent.update(batch) draws from agent.rng, with the reference's key derivation.  Returns (new_rng,
    noise dict)."""
    new_rng, keys = fql_update_keys(rng, partitionable)
    return new_rng, noise_from_keys(keys, batch_size, action_dim, partitionable)


def fql_total_loss_noise(rng, batch_size: int, action_dim: int, partitionable=None):
    """Noise of FQLAgent.total_loss(batch, grad_params, rng) (agents/fql.py:94-111): rng itself is the argument (or
    agent.rng when None, as the validation probe main.py:284 does)."""
    return noise_from_keys(fql_total_loss_keys(rng, partitionable), batch_size, action_dim, partitionable)


def sample_actions_noise(seed, lead_shape, action_dim: int, partitionable=None) -> np.ndarray:
    """The noise FQLAgent.sample_actions(obs, seed) draws: normal(split(seed)[0], (*obs.shape[:-1], action_dim))."""
    return normal(split(seed, 2, partitionable)[0], tuple(lead_shape) + (action_dim,), partitionable)


def erfinv_f32(x) -> np.ndarray:
    """Single-precision inverse error function as XLA evaluates lax.erf_inv for float32 (M. Giles, "Approximating the erfinv
    function", 2010: two polynomial branches in w = -log((1 - x)(1 + x))), in float32 arithmetic.  The device generator uses the same
    coefficients; exposed for its test."""
    x = np.asarray(x, dtype=np.float32)
    w = -np.log((np.float32(1) - x) * (np.float32(1) + x)).astype(np.float32)
    lt = w < np.float32(5)
    wa = np.where(lt, w - np.float32(2.5), np.sqrt(np.maximum(w, 0)).astype(np.float32) - np.float32(3)).astype(np.float32)
    ca = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941)
    cb = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682)
    p = np.where(lt, np.float32(ca[0]), np.float32(cb[0])).astype(np.float32)
    for a, b in zip(ca[1:], cb[1:]):
        p = (np.where(lt, np.float32(a), np.float32(b)) + p * wa).astype(np.float32)
    return (p * x).astype(np.float32)
