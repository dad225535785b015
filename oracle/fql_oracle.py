"""CPU oracle: numpy restatement of the reference FQL gradient step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fql_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, as
the checker / the timed CPU baseline -- never as the product path.

PARITY UNPINNED at the JAX boundary: the reference (``/root/reference``) is pure
Python/JAX, ships no tests, fixtures or golden vectors, and jax/flax/optax are absent from
this image.  This restatement is pinned instead by
(1) an independent torch-autograd restatement (``fql_oracle_torch.py``),
(2) central finite differences in float64, and
(3) semantic known-answer tests for every reference subtlety listed below
(see ``tests/test_oracle.py``).  Every comparison made with it is "vs CPU restatement of
the reference", never "vs JAX".

What each function follows (paths relative to /root/reference):

* ``gelu_tanh``          utils/networks.py:46   (flax ``nn.gelu`` == tanh approximation)
* ``layer_norm``         utils/networks.py:58   (flax LayerNorm: eps 1e-6, fast variance
                                                 max(0, E[x^2]-E[x]^2), scale+bias)
* ``mlp_forward``        utils/networks.py:53-60 (Dense -> GELU -> LN, none after last Dense)
* ``Value`` ensemble     utils/networks.py:14-24,172-195 (two members, params stacked on axis 0)
* ``ActorVectorField``   utils/networks.py:225-235 (concat(obs, act[, t]))
* ``critic_loss``        agents/fql.py:22-44
* ``actor_loss``         agents/fql.py:46-92
* ``total_loss``         agents/fql.py:94-111
* ``target_update``      agents/fql.py:113-120  (Polyak from the PRE-step critic, F4)
* ``update``             agents/fql.py:122-133
* ``sample_actions``     agents/fql.py:135-153
* ``compute_flow_actions`` agents/fql.py:155-171
* ``create``/init        agents/fql.py:173-246, utils/networks.py:9-11 (Glorot uniform)
* grad stats + Adam      utils/flax_utils.py:120-159, agents/fql.py:237 (optax.adam defaults)

Noise is an explicit input (PRNG row): the five tensors the reference
draws inside one ``update`` are

    eps1 [B,A]  critic_loss  sample_actions(next_obs) noise      agents/fql.py:25,143-150
    x0   [B,A]  actor_loss   x_0                                  agents/fql.py:52
    t    [B,1]  actor_loss   t ~ U[0,1)                           agents/fql.py:54
    z    [B,A]  actor_loss   distillation noises                 agents/fql.py:63
    eps2 [B,A]  actor_loss   sample_actions(obs) noise (metric)  agents/fql.py:82
"""
from __future__ import annotations

import copy
import math
from typing import Dict, Tuple

import numpy as np

from oracle import encoder_oracle as ENC

SQRT_2_OVER_PI = math.sqrt(2.0 / math.pi)
GELU_C = 0.044715
LN_EPS = 1e-6

INFO_KEYS = (
    'critic/critic_loss', 'critic/q_mean', 'critic/q_max', 'critic/q_min',
    'actor/actor_loss', 'actor/bc_flow_loss', 'actor/distill_loss', 'actor/q_loss',
    'actor/q', 'actor/mse', 'grad/max', 'grad/min', 'grad/norm',
)

NOISE_KEYS = ('eps1', 'x0', 't', 'z', 'eps2')

MODULES = ('modules_critic', 'modules_target_critic', 'modules_actor_bc_flow',
           'modules_actor_onestep_flow')


def get_config() -> dict:
    """Defaults of agents/fql.py:249-270 as a plain dict."""
    return dict(
        agent_name='fql', ob_dims=None, action_dim=None, lr=3e-4, batch_size=256,
        actor_hidden_dims=(512, 512, 512, 512), value_hidden_dims=(512, 512, 512, 512),
        layer_norm=True, actor_layer_norm=False, discount=0.99, tau=0.005, q_agg='mean',
        alpha=300.0, flow_steps=10, normalize_q_loss=False, encoder=None,
    )


# ----------------------------------------------------------------------------------------
# elementary ops (forward + hand-derived backward)
# ----------------------------------------------------------------------------------------
def gelu_tanh(x):
    u = SQRT_2_OVER_PI * (x + GELU_C * x * x * x)
    return 0.5 * x * (1.0 + np.tanh(u))


def gelu_tanh_grad(x):
    u = SQRT_2_OVER_PI * (x + GELU_C * x * x * x)
    th = np.tanh(u)
    du = SQRT_2_OVER_PI * (1.0 + 3.0 * GELU_C * x * x)
    return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * du


def layer_norm(x, scale, bias):
    """Returns (y, xhat, rstd). Stats over the last axis, fast variance clamped at 0."""
    mean = x.mean(axis=-1, keepdims=True)
    mean2 = (x * x).mean(axis=-1, keepdims=True)
    var = np.maximum(0.0, mean2 - mean * mean)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(LN_EPS))
    xhat = (x - mean) * rstd
    return xhat * scale + bias, xhat, rstd


def layer_norm_bwd(dy, xhat, rstd, scale):
    """Returns (dx, dscale, dbias)."""
    dxhat = dy * scale
    m1 = dxhat.mean(axis=-1, keepdims=True)
    m2 = (dxhat * xhat).mean(axis=-1, keepdims=True)
    dx = rstd * (dxhat - m1 - xhat * m2)
    return dx, (dy * xhat).sum(axis=0), dy.sum(axis=0)


def _n_layers(net: dict) -> int:
    return sum(1 for k in net if k.startswith('Dense_'))


def mlp_forward(net: dict, x, member=None, keep=False):
    """MLP of utils/networks.py:34-61.  ``member`` selects an ensemble slice of stacked leaves."""
    def leaf(a):
        return a if member is None else a[member]

    n = _n_layers(net)
    cache = []
    for i in range(n):
        d = net[f'Dense_{i}']
        z = x @ leaf(d['kernel']) + leaf(d['bias'])
        if i + 1 < n:
            g = gelu_tanh(z)
            ln = net.get(f'LayerNorm_{i}')
            if ln is not None:
                y, xhat, rstd = layer_norm(g, leaf(ln['scale']), leaf(ln['bias']))
            else:
                y, xhat, rstd = g, None, None
            if keep:
                cache.append((x, z, xhat, rstd))
            x = y
        else:
            if keep:
                cache.append((x, None, None, None))
            x = z
    return (x, cache) if keep else x


def mlp_backward(net: dict, cache, dout, member=None, want_param_grads=True):
    """Backward of ``mlp_forward``.  Returns (dx_input, grads-with-the-same-tree-as-net or None)."""
    def leaf(a):
        return a if member is None else a[member]

    n = _n_layers(net)
    grads = {} if want_param_grads else None
    dy = dout
    for i in reversed(range(n)):
        x, z, xhat, rstd = cache[i]
        if i + 1 < n:
            ln = net.get(f'LayerNorm_{i}')
            if ln is not None:
                dg, dscale, dbias = layer_norm_bwd(dy, xhat, rstd, leaf(ln['scale']))
                if want_param_grads:
                    grads[f'LayerNorm_{i}'] = {'scale': dscale, 'bias': dbias}
            else:
                dg = dy
            dz = dg * gelu_tanh_grad(z)
        else:
            dz = dy
        if want_param_grads:
            grads[f'Dense_{i}'] = {'kernel': x.T @ dz, 'bias': dz.sum(axis=0)}
        dy = dz @ leaf(net[f'Dense_{i}']['kernel']).T
    return dy, grads


# ----------------------------------------------------------------------------------------
# parameter trees
# ----------------------------------------------------------------------------------------
def glorot_uniform(rng: np.random.Generator, fan_in, fan_out, shape, dtype):
    """utils/networks.py:9-11: variance_scaling(1.0, 'fan_avg', 'uniform')."""
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_params(seed: int, obs_dim: int, act_dim: int, config: dict, dtype=np.float32) -> dict:
    """Parameter tree with the reference's leaf names and shapes (agents/fql.py:205-242).

    Values come from numpy's default_rng (JAX's threefry init is not reproducible here:
    SURVEY.md 8c PRNG row), distributions match: Glorot-uniform kernels, zero biases,
    LN scale 1 / bias 0; target critic := critic (agents/fql.py:241-242).
    """
    rng = np.random.default_rng(seed)
    vh = tuple(config['value_hidden_dims'])
    ah = tuple(config['actor_hidden_dims'])
    enc_name = config.get('encoder')
    ob_shape = None
    if enc_name is not None:   # agents/fql.py:196-202: obs_dim is the image shape (H, W, C); the MLPs see the encoding
        ob_shape = tuple(obs_dim)
        obs_dim = ENC.encoder_out_dim(enc_name)

    def actor(in_dim, ln):
        dims = (in_dim,) + ah + (act_dim,)
        net = {}
        for i in range(len(dims) - 1):
            net[f'Dense_{i}'] = {
                'kernel': glorot_uniform(rng, dims[i], dims[i + 1], (dims[i], dims[i + 1]), dtype),
                'bias': np.zeros((dims[i + 1],), dtype),
            }
            if ln and i + 1 < len(dims) - 1:
                net[f'LayerNorm_{i}'] = {'scale': np.ones((dims[i + 1],), dtype),
                                         'bias': np.zeros((dims[i + 1],), dtype)}
        return {'mlp': net}

    def critic():
        dims = (obs_dim + act_dim,) + vh + (1,)
        net = {}
        for i in range(len(dims) - 1):
            net[f'Dense_{i}'] = {
                'kernel': glorot_uniform(rng, dims[i], dims[i + 1], (2, dims[i], dims[i + 1]), dtype),
                'bias': np.zeros((2, dims[i + 1]), dtype),
            }
            if config['layer_norm'] and i + 1 < len(dims) - 1:
                net[f'LayerNorm_{i}'] = {'scale': np.ones((2, dims[i + 1]), dtype),
                                         'bias': np.zeros((2, dims[i + 1]), dtype)}
        return {'value_net': net}

    params = {
        'modules_critic': critic(),
        'modules_actor_bc_flow': actor(obs_dim + act_dim + 1, config['actor_layer_norm']),
        'modules_actor_onestep_flow': actor(obs_dim + act_dim, config['actor_layer_norm']),
    }
    if enc_name is not None:
        # one encoder per module (agents/fql.py:199-202).  `actor_bc_flow_encoder` (agents/fql.py:230-232) is the SAME
        # module instance as actor_bc_flow's encoder; it is restated here as sharing that parameter set (author intent per
        # the comment there; whether flax stores it once or twice could not be checked: SURVEY.md 8f N1).
        for m in ('modules_critic', 'modules_actor_bc_flow', 'modules_actor_onestep_flow'):
            params[m]['encoder'] = ENC.init_encoder_params(rng, ob_shape, enc_name, dtype)
    params['modules_target_critic'] = copy.deepcopy(params['modules_critic'])
    return params


def tree_leaves_with_path(tree, prefix=''):
    """Sorted-key traversal = jax.tree_util order for dicts."""
    out = []
    for k in sorted(tree):
        v = tree[k]
        p = f'{prefix}/{k}' if prefix else k
        if isinstance(v, dict):
            out.extend(tree_leaves_with_path(v, p))
        else:
            out.append((p, v))
    return out


def tree_map(f, *trees):
    t0 = trees[0]
    if isinstance(t0, dict):
        return {k: tree_map(f, *[t[k] for t in trees]) for k in t0}
    return f(*trees)


# ----------------------------------------------------------------------------------------
# the agent
# ----------------------------------------------------------------------------------------
class OracleFQL:
    """Restatement of FQLAgent (agents/fql.py:15-246) with explicit noise."""

    def __init__(self, params: dict, config: dict, obs_dim: int, act_dim: int, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.config = dict(config)
        self.enc_name = self.config.get('encoder')
        self.ob_shape = tuple(obs_dim) if self.enc_name is not None else None
        self.obs_dim = ENC.encoder_out_dim(self.enc_name) if self.enc_name is not None else obs_dim
        self.act_dim = act_dim
        self.params = tree_map(lambda a: np.array(a, dtype=self.dtype), params)
        # optax.adam state: count=0, mu=0, nu=0 (agents/fql.py:237, utils/flax_utils.py:74-76)
        self.mu = tree_map(np.zeros_like, self.params)
        self.nu = tree_map(np.zeros_like, self.params)
        self.count = 0
        self.step = 1  # utils/flax_utils.py:81
        # test hook: when set, every use the reference makes with params=None (stored params, constants under jax.grad:
        # utils/flax_utils.py:90-118) reads this tree instead of self.params, so finite differences of total_loss w.r.t.
        # self.params see exactly the dependence jax.grad differentiates
        self.frozen = None

    def _P(self, stored=False):
        return self.frozen if (stored and self.frozen is not None) else self.params

    @classmethod
    def create(cls, seed, obs_dim, act_dim, config, dtype=np.float32):
        return cls(init_params(seed, obs_dim, act_dim, config, dtype), config, obs_dim, act_dim, dtype)

    # -- network helpers ------------------------------------------------------------
    def _c(self, x):
        return np.asarray(x, dtype=self.dtype)

    def _enc(self, name, obs, keep=False, stored=False):
        """Module `name`'s encoder applied to raw observations (identity for state-based agents)."""
        if self.enc_name is None:
            return (self._c(obs), None) if keep else self._c(obs)
        return ENC.impala_forward(self._P(stored)[name]['encoder'], obs, keep=keep, dtype=self.dtype.type)

    def _actor(self, name, obs, act, t=None, keep=False, stored=False):
        """`obs` is already encoded (utils/networks.py:221-222 with the encoder applied by the caller)."""
        xs = [obs, act] if t is None else [obs, act, t]
        x = np.concatenate(xs, axis=-1)
        return mlp_forward(self._P(stored)[name]['mlp'], x, keep=keep)

    def _critic(self, name, obs, act, keep=False, stored=False):
        x = np.concatenate([obs, act], axis=-1)
        net = self._P(stored)[name]['value_net']
        outs, caches = [], []
        for e in range(2):
            r = mlp_forward(net, x, member=e, keep=keep)
            if keep:
                outs.append(r[0][:, 0]); caches.append(r[1])
            else:
                outs.append(r[:, 0])
        q = np.stack(outs, axis=0)  # [2, B]
        return (q, caches) if keep else q

    # -- public API -----------------------------------------------------------------
    def sample_actions(self, observations, noises):
        """agents/fql.py:135-153 with the normal draw passed in."""
        a = self._actor('modules_actor_onestep_flow', self._enc('modules_actor_onestep_flow', observations, stored=True),
                        self._c(noises), stored=True)
        return np.clip(a, -1, 1)

    def compute_flow_actions(self, observations, noises):
        """agents/fql.py:155-171."""
        obs = self._enc('modules_actor_bc_flow', observations, stored=True)  # encoded once (agents/fql.py:162-163, is_encoded=True :168)
        a = self._c(noises)
        n = int(self.config['flow_steps'])
        for i in range(n):
            t = np.full(obs.shape[:-1] + (1,), i / n, dtype=self.dtype)
            v = self._actor('modules_actor_bc_flow', obs, a, t, stored=True)
            a = a + v / self.dtype.type(n)
        return np.clip(a, -1, 1)

    def _losses(self, batch, noise, want_grads: bool):
        cfg = self.config
        dt = self.dtype.type
        vis = self.enc_name is not None
        raw_obs, raw_nobs = batch['observations'], batch['next_observations']
        act = self._c(batch['actions'])
        rew = self._c(batch['rewards']).reshape(-1); mask = self._c(batch['masks']).reshape(-1)
        eps1, x0, z, eps2 = (self._c(noise[k]) for k in ('eps1', 'x0', 'z', 'eps2'))
        t = self._c(noise['t']).reshape(-1, 1)
        B, A = act.shape
        info = {}

        # ---- critic loss (agents/fql.py:22-44)
        next_actions = np.clip(self.sample_actions(raw_nobs, eps1), -1, 1)
        next_qs = self._critic('modules_target_critic', self._enc('modules_target_critic', raw_nobs, stored=True), next_actions,
                               stored=True)
        next_q = next_qs.min(axis=0) if cfg['q_agg'] == 'min' else next_qs.mean(axis=0)
        target_q = rew + dt(cfg['discount']) * mask * next_q
        obs_c, enc_c_cache = self._enc('modules_critic', raw_obs, keep=True)
        q, c_caches = self._critic('modules_critic', obs_c, act, keep=True)
        critic_loss = np.square(q - target_q).mean()
        info['critic/critic_loss'] = critic_loss
        info['critic/q_mean'] = q.mean(); info['critic/q_max'] = q.max(); info['critic/q_min'] = q.min()

        # ---- actor loss (agents/fql.py:46-92)
        x_t = (1 - t) * x0 + t * act
        vel = act - x0
        obs_bc, enc_bc_cache = self._enc('modules_actor_bc_flow', raw_obs, keep=True)
        pred, bc_cache = self._actor('modules_actor_bc_flow', obs_bc, x_t, t, keep=True)
        bc_flow_loss = np.mean((pred - vel) ** 2)

        target_flow_actions = self.compute_flow_actions(raw_obs, z)
        obs_os, enc_os_cache = self._enc('modules_actor_onestep_flow', raw_obs, keep=True)
        a_raw, os_cache = self._actor('modules_actor_onestep_flow', obs_os, z, keep=True)
        distill_loss = np.mean((a_raw - target_flow_actions) ** 2)

        a_clip = np.clip(a_raw, -1, 1)
        # agents/fql.py:70: critic called with params=None -> stored params (same encoder, same obs: same encoding as obs_c)
        obs_cs = obs_c if self.frozen is None else self._enc('modules_critic', raw_obs, stored=True)
        qs, q_caches = self._critic('modules_critic', obs_cs, a_clip, keep=True, stored=True)
        qm = qs.mean(axis=0)
        q_loss = -qm.mean()
        lam = dt(1.0)
        if cfg['normalize_q_loss']:
            lam = dt(1.0) / np.abs(qm).mean()
            q_loss = lam * q_loss
        actor_loss = bc_flow_loss + dt(cfg['alpha']) * distill_loss + q_loss

        actions = self.sample_actions(raw_obs, eps2)
        mse = np.mean((actions - act) ** 2)
        info.update({'actor/actor_loss': actor_loss, 'actor/bc_flow_loss': bc_flow_loss,
                     'actor/distill_loss': distill_loss, 'actor/q_loss': q_loss,
                     'actor/q': qm.mean(), 'actor/mse': mse})
        loss = critic_loss + actor_loss
        if not want_grads:
            return loss, info, None

        # ---- backward (what jax.grad at utils/flax_utils.py:137 produces)
        grads = tree_map(np.zeros_like, self.params)  # target-critic leaves stay zero (F5)

        # critic params <- critic_loss.  mean over 2B elements (agents/fql.py:37)
        cnet = self.params['modules_critic']['value_net']
        gc = grads['modules_critic']['value_net']
        dq = (2.0 / (2 * B)) * (q - target_q)  # [2,B]
        d_enc = 0
        for e in range(2):
            dx, g = mlp_backward(cnet, c_caches[e], dq[e][:, None].astype(self.dtype), member=e)
            d_enc = d_enc + dx[:, :self.obs_dim]
            for ln, sub in g.items():
                for k, v in sub.items():
                    gc[ln][k][e] = v
        if vis:  # the critic's encoder sees only the critic loss (the actor loss calls the critic with constant params)
            grads['modules_critic']['encoder'] = ENC.impala_backward(self.params['modules_critic']['encoder'], enc_c_cache, d_enc)

        # bc_flow params <- bc_flow_loss
        dpred = (2.0 / (B * A)) * (pred - vel)
        dx, g = mlp_backward(self.params['modules_actor_bc_flow']['mlp'], bc_cache, dpred.astype(self.dtype))
        grads['modules_actor_bc_flow']['mlp'] = g
        if vis:  # the distillation target is computed outside the gradient (agents/fql.py:64: no params passed)
            grads['modules_actor_bc_flow']['encoder'] = ENC.impala_backward(
                self.params['modules_actor_bc_flow']['encoder'], enc_bc_cache, dx[:, :self.obs_dim])

        # onestep params <- alpha*distill + q_loss (critic params constant: params=None, flax_utils.py:90-118)
        da = dt(cfg['alpha']) * (2.0 / (B * A)) * (a_raw - target_flow_actions)
        dact = np.zeros_like(a_raw)
        for e in range(2):
            dqe = np.full((B, 1), -lam / (2 * B), dtype=self.dtype)
            dx, _ = mlp_backward(self._P(True)['modules_critic']['value_net'], q_caches[e], dqe, member=e, want_param_grads=False)
            dact += dx[:, self.obs_dim:self.obs_dim + A]
        inside = (a_raw > -1) & (a_raw < 1)
        da = da + dact * inside
        dx, g = mlp_backward(self.params['modules_actor_onestep_flow']['mlp'], os_cache, da.astype(self.dtype))
        grads['modules_actor_onestep_flow']['mlp'] = g
        if vis:
            grads['modules_actor_onestep_flow']['encoder'] = ENC.impala_backward(
                self.params['modules_actor_onestep_flow']['encoder'], enc_os_cache, dx[:, :self.obs_dim])
        return loss, info, grads

    def total_loss(self, batch, noise):
        """agents/fql.py:94-111 with grad_params=None (validation probe, main.py:284)."""
        loss, info, _ = self._losses(batch, noise, want_grads=False)
        return loss, info

    def grads(self, batch, noise):
        return self._losses(batch, noise, want_grads=True)

    @staticmethod
    def grad_stats(grads) -> Dict[str, float]:
        """utils/flax_utils.py:139-157: max, min over all leaves; norm = SUM of per-leaf L2 norms."""
        leaves = [g for _, g in tree_leaves_with_path(grads)]
        return {
            'grad/max': max(float(g.max()) for g in leaves),
            'grad/min': min(float(g.min()) for g in leaves),
            'grad/norm': float(sum(np.sqrt(np.sum(np.square(g.astype(np.float64)))) for g in leaves)),
        }

    def apply_gradients(self, grads):
        """optax.adam(lr) + apply_updates (utils/flax_utils.py:120-130), then Polyak from OLD critic."""
        cfg = self.config
        dt = self.dtype.type
        b1, b2, eps, lr = dt(0.9), dt(0.999), dt(1e-8), dt(cfg['lr'])
        self.count += 1
        c1 = dt(1.0 - 0.9 ** self.count); c2 = dt(1.0 - 0.999 ** self.count)
        old_critic = copy.deepcopy(self.params['modules_critic'])
        old_target = self.params['modules_target_critic']

        def upd(p, g, m, v):
            m[...] = b1 * m + (dt(1) - b1) * g
            v[...] = b2 * v + (dt(1) - b2) * g * g
            mhat = m / c1
            vhat = v / c2
            return p - lr * mhat / (np.sqrt(vhat) + eps)

        self.params = tree_map(upd, self.params, grads, self.mu, self.nu)
        tau = dt(cfg['tau'])
        # agents/fql.py:113-120: reads self.network.params (pre-step) for BOTH critic and target
        self.params['modules_target_critic'] = tree_map(
            lambda p, tp: p * tau + tp * (dt(1) - tau), old_critic, old_target)
        self.step += 1

    def update(self, batch, noise) -> Tuple[float, dict]:
        """agents/fql.py:122-133."""
        loss, info, grads = self._losses(batch, noise, want_grads=True)
        info.update(self.grad_stats(grads))
        self.apply_gradients(grads)
        return loss, {k: float(info[k]) for k in INFO_KEYS}


# ----------------------------------------------------------------------------------------
# synthetic data -- shared by tests and bench
# ----------------------------------------------------------------------------------------
def make_synthetic_dataset(n: int, obs_dim: int, act_dim: int, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    obs = rng.standard_normal((n, obs_dim)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    act = np.clip(act, -1 + 1e-5, 1 - 1e-5).astype(np.float32)  # envs/env_utils.py:138-146
    rew = np.where(rng.uniform(size=n) < 0.99, -1.0, 0.0).astype(np.float32)
    masks = np.where(rng.uniform(size=n) < 0.99, 1.0, 0.0).astype(np.float32)
    nobs = (obs + 0.1 * rng.standard_normal((n, obs_dim))).astype(np.float32)
    return dict(observations=obs, actions=act, rewards=rew, masks=masks,
                next_observations=nobs, terminals=(1.0 - masks).astype(np.float32))


def make_noise(batch_size: int, act_dim: int, seed: int) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    return dict(
        eps1=rng.standard_normal((batch_size, act_dim)).astype(np.float32),
        x0=rng.standard_normal((batch_size, act_dim)).astype(np.float32),
        t=rng.uniform(size=(batch_size, 1)).astype(np.float32),
        z=rng.standard_normal((batch_size, act_dim)).astype(np.float32),
        eps2=rng.standard_normal((batch_size, act_dim)).astype(np.float32),
    )


def sample_batch(ds: Dict[str, np.ndarray], idxs) -> Dict[str, np.ndarray]:
    """utils/datasets.py:94-100: arr[idxs] per key."""
    return {k: v[idxs] for k, v in ds.items()}
