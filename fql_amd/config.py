"""FQL config with the reference's keys and defaults (agents/fql.py:249-270)."""


class ConfigDict(dict):
    """dict with attribute access (the subset of ml_collections.ConfigDict the call sites use)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def get_config():
    return ConfigDict(
        agent_name='fql',
        ob_dims=None,  # set automatically by create()
        action_dim=None,  # set automatically by create()
        lr=3e-4,
        batch_size=256,
        actor_hidden_dims=(512, 512, 512, 512),
        value_hidden_dims=(512, 512, 512, 512),
        layer_norm=True,
        actor_layer_norm=False,
        discount=0.99,
        tau=0.005,
        q_agg='mean',
        alpha=300.0,
        flow_steps=10,
        normalize_q_loss=False,
        encoder=None,
        rng='engine',  # not a reference key: 'engine' = device Philox noise; 'jax' / 'jax_partitionable' = the reference's key derivation with
                       # JAX's original / partitionable threefry layout (keys on the host, tensors on the device: fql_noise_from_jax_keys)
        precision='fp32',  # not a reference key: 'fp32' = fp32 matrix cores (exact fma chains); 'bf16x3' = split-bf16 products on the bf16 matrix
                           # cores with fp32 accumulation (fql_config.precision = 2; ~1e-5 relative on the dense products)
        rng_device=True,  # not a reference key: False draws the JAX-mode tensors on the host (fql_amd/jax_prng.py) and ships them H2D
    )
