"""CPU oracle #2: torch-autograd restatement of the reference FQL loss + update.

TEST INFRASTRUCTURE ONLY (same rule as ``fql_oracle.py``: never imported by ``fql_amd/``).
PARITY UNPINNED at the JAX boundary (see ``fql_oracle.py`` header).

Written independently of ``fql_oracle.py``'s hand-derived backward: gradients come from
``torch.autograd`` applied to a direct transcription of agents/fql.py:22-111, the
"params=None => constant" rule of utils/flax_utils.py:90-118 is expressed with
``.detach()`` on the parameter trees, Adam is a transcription of optax.adam's update rule,
and grad statistics follow utils/flax_utils.py:139-157.  ``tests/test_oracle.py`` checks
numpy-oracle == torch-oracle; ``bench.py`` times this module as the CPU baseline
("CPU restatement of the reference path (not JAX)").
"""
from __future__ import annotations

import math

import torch

from .fql_oracle import INFO_KEYS, tree_leaves_with_path


def _to_torch(tree, dtype):
    if isinstance(tree, dict):
        return {k: _to_torch(v, dtype) for k, v in tree.items()}
    return torch.tensor(tree, dtype=dtype)


def _map(f, *trees):
    if isinstance(trees[0], dict):
        return {k: _map(f, *[t[k] for t in trees]) for k in trees[0]}
    return f(*trees)


def _gelu(x):
    # utils/networks.py:46 -> flax nn.gelu == tanh approximation
    return torch.nn.functional.gelu(x, approximate='tanh')


def _layer_norm(x, scale, bias):
    # flax LayerNorm defaults: eps=1e-6, use_fast_variance=True
    mean = x.mean(-1, keepdim=True)
    var = torch.clamp((x * x).mean(-1, keepdim=True) - mean * mean, min=0.0)
    return (x - mean) * torch.rsqrt(var + 1e-6) * scale + bias


def _mlp(net, x, member=None):
    n = sum(1 for k in net if k.startswith('Dense_'))
    pick = (lambda a: a) if member is None else (lambda a: a[member])
    for i in range(n):
        x = x @ pick(net[f'Dense_{i}']['kernel']) + pick(net[f'Dense_{i}']['bias'])
        if i + 1 < n:
            x = _gelu(x)
            if f'LayerNorm_{i}' in net:
                x = _layer_norm(x, pick(net[f'LayerNorm_{i}']['scale']), pick(net[f'LayerNorm_{i}']['bias']))
    return x


def _encode(mod, obs):
    """utils/encoders.py:83-100 (ImpalaEncoder) when the module has an encoder, identity otherwise.  Written with
    torch.nn.functional ops, independently of oracle/encoder_oracle.py's im2col restatement."""
    if 'encoder' not in mod:
        return obs
    F = torch.nn.functional
    p = mod['encoder']
    x = (obs / 255.0).permute(0, 3, 1, 2)

    def conv(x, c):   # flax Conv, SAME, HWIO kernel -> torch OIHW
        return F.conv2d(x, c['kernel'].permute(3, 2, 0, 1), c['bias'], padding=1)

    ns = sum(1 for k in p if k.startswith('stack_blocks_'))
    for si in range(ns):
        st = p[f'stack_blocks_{si}']
        x = conv(x, st['Conv_0'])
        x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), value=float('-inf')), 3, 2)   # SAME: the pad sits at the end
        for b in range((len(st) - 1) // 2):
            inp = x
            x = conv(F.relu(x), st[f'Conv_{1 + 2 * b}'])
            x = conv(F.relu(x), st[f'Conv_{2 + 2 * b}'])
            x = x + inp
    x = F.relu(x).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    for i in range(len(p['MLP_0'])):
        d = p['MLP_0'][f'Dense_{i}']
        x = _gelu(x @ d['kernel'] + d['bias'])
    return x


def _value(mod, obs, act):
    x = torch.cat([_encode(mod, obs), act], -1)
    return torch.stack([_mlp(mod['value_net'], x, e).squeeze(-1) for e in range(2)], 0)


def _vf(mod, obs, act, t=None, encoded=False):
    if not encoded:
        obs = _encode(mod, obs)
    return _mlp(mod['mlp'], torch.cat([obs, act] if t is None else [obs, act, t], -1))


class TorchFQL:
    def __init__(self, params, config, dtype=torch.float32):
        self.dtype = dtype
        self.config = dict(config)
        self.params = _to_torch(params, dtype)
        self.mu = _map(torch.zeros_like, self.params)
        self.nu = _map(torch.zeros_like, self.params)
        self.count = 0

    def _t(self, a):
        return torch.as_tensor(a, dtype=self.dtype)

    def total_loss(self, batch, noise, grad_params=None):
        """agents/fql.py:94-111.  ``grad_params`` None => everything uses stored params."""
        cfg = self.config
        stored = _map(lambda p: p.detach(), self.params)
        gp = stored if grad_params is None else grad_params
        obs, act, nobs = self._t(batch['observations']), self._t(batch['actions']), self._t(batch['next_observations'])
        rew, mask = self._t(batch['rewards']).reshape(-1), self._t(batch['masks']).reshape(-1)
        eps1, x0, z, eps2 = (self._t(noise[k]) for k in ('eps1', 'x0', 'z', 'eps2'))
        t = self._t(noise['t']).reshape(-1, 1)

        def sample_actions(o, n):  # agents/fql.py:135-153
            return torch.clamp(_vf(stored['modules_actor_onestep_flow'], o, n), -1, 1)

        # critic_loss, agents/fql.py:22-44
        na = torch.clamp(sample_actions(nobs, eps1), -1, 1)
        nqs = _value(stored['modules_target_critic'], nobs, na)
        nq = nqs.min(0).values if cfg['q_agg'] == 'min' else nqs.mean(0)
        target_q = rew + cfg['discount'] * mask * nq
        q = _value(gp['modules_critic'], obs, act)
        critic_loss = ((q - target_q) ** 2).mean()

        # actor_loss, agents/fql.py:46-92
        x_t = (1 - t) * x0 + t * act
        vel = act - x0
        pred = _vf(gp['modules_actor_bc_flow'], obs, x_t, t)
        bc = ((pred - vel) ** 2).mean()
        a = z
        n = int(cfg['flow_steps'])
        eobs = _encode(stored['modules_actor_bc_flow'], obs)   # agents/fql.py:162-163: encoded once, is_encoded=True
        for i in range(n):  # agents/fql.py:166-169
            ti = torch.full((obs.shape[0], 1), i / n, dtype=self.dtype)
            a = a + _vf(stored['modules_actor_bc_flow'], eobs, a, ti, encoded=True) / n
        tgt = torch.clamp(a, -1, 1)
        aa = _vf(gp['modules_actor_onestep_flow'], obs, z)
        distill = ((aa - tgt) ** 2).mean()
        qs = _value(stored['modules_critic'], obs, torch.clamp(aa, -1, 1))
        qm = qs.mean(0)
        q_loss = -qm.mean()
        if cfg['normalize_q_loss']:
            q_loss = (1 / qm.abs().mean()).detach() * q_loss
        actor_loss = bc + cfg['alpha'] * distill + q_loss
        mse = ((sample_actions(obs, eps2) - act) ** 2).mean()

        info = {
            'critic/critic_loss': critic_loss, 'critic/q_mean': q.mean(), 'critic/q_max': q.max(),
            'critic/q_min': q.min(), 'actor/actor_loss': actor_loss, 'actor/bc_flow_loss': bc,
            'actor/distill_loss': distill, 'actor/q_loss': q_loss, 'actor/q': qm.mean(), 'actor/mse': mse,
        }
        return critic_loss + actor_loss, info

    def grads(self, batch, noise):
        gp = _map(lambda p: p.detach().clone().requires_grad_(True), self.params)
        loss, info = self.total_loss(batch, noise, grad_params=gp)
        loss.backward()
        grads = _map(lambda p: torch.zeros_like(p) if p.grad is None else p.grad, gp)
        return loss.detach(), info, grads

    def update(self, batch, noise):
        """agents/fql.py:122-133 + utils/flax_utils.py:120-159."""
        cfg = self.config
        loss, info, grads = self.grads(batch, noise)
        leaves = [g for _, g in tree_leaves_with_path(grads)]
        info = {k: float(v.detach()) for k, v in info.items()}
        info['grad/max'] = max(float(g.max()) for g in leaves)
        info['grad/min'] = min(float(g.min()) for g in leaves)
        info['grad/norm'] = float(sum(torch.linalg.norm(g.reshape(-1)) for g in leaves))

        self.count += 1
        c1, c2 = 1 - 0.9 ** self.count, 1 - 0.999 ** self.count
        old_c = _map(lambda p: p.clone(), self.params['modules_critic'])
        old_t = self.params['modules_target_critic']

        def adam(p, g, m, v):
            m.mul_(0.9).add_(g, alpha=0.1)
            v.mul_(0.999).addcmul_(g, g, value=0.001)
            return p - cfg['lr'] * (m / c1) / (torch.sqrt(v / c2) + 1e-8)

        with torch.no_grad():
            self.params = _map(adam, self.params, grads, self.mu, self.nu)
            tau = cfg['tau']
            self.params['modules_target_critic'] = _map(lambda p, tp: p * tau + tp * (1 - tau), old_c, old_t)
        return float(loss), {k: info[k] for k in INFO_KEYS}
