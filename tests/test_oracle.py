"""Pins the CPU oracle (oracle/fql_oracle.py): numpy hand-derived backward == torch autograd,
finite differences, and known-answer tests for the reference subtleties SURVEY.md F2-F5 lists.
All comparisons are "vs CPU restatement of the reference" (parity unpinned at the JAX boundary)."""
import copy

import numpy as np
import pytest
import torch

from oracle import fql_oracle as O
from oracle.fql_oracle_torch import TorchFQL


def small_cfg(**kw):
    cfg = O.get_config()
    cfg.update(actor_hidden_dims=(32, 32, 32, 32), value_hidden_dims=(32, 32, 32, 32), alpha=10.0)
    cfg.update(kw)
    return cfg


def make_case(B=16, obs_dim=7, act_dim=3, seed=0, dtype=np.float64, **kw):
    cfg = small_cfg(**kw)
    params = O.init_params(seed, obs_dim, act_dim, cfg, dtype)
    # non-trivial biases / LN params so every leaf matters
    rng = np.random.default_rng(seed + 100)
    for _, leaf in O.tree_leaves_with_path(params):
        if leaf.ndim <= 2 and leaf.shape[-1] != 0 and np.all((leaf == 0) | (leaf == 1)):
            leaf += 0.1 * rng.standard_normal(leaf.shape)
    params['modules_target_critic'] = O.tree_map(
        lambda a: a + 0.01 * rng.standard_normal(a.shape), params['modules_critic'])
    ds = O.make_synthetic_dataset(64, obs_dim, act_dim, seed=seed)
    batch = O.sample_batch(ds, np.arange(B))
    noise = O.make_noise(B, act_dim, seed + 7)
    return cfg, params, batch, noise


@pytest.mark.parametrize('q_agg,norm', [('mean', False), ('min', False), ('mean', True)])
def test_numpy_matches_torch_autograd_fp64(q_agg, norm):
    cfg, params, batch, noise = make_case(q_agg=q_agg, normalize_q_loss=norm)
    a = O.OracleFQL(params, cfg, 7, 3, np.float64)
    b = TorchFQL(params, cfg, torch.float64)
    la, ia, ga = a.grads(batch, noise)
    lb, ib, gb = b.grads(batch, noise)
    assert abs(float(la) - float(lb)) < 1e-11
    for k in ia:
        assert abs(float(ia[k]) - float(ib[k].detach())) < 1e-11, k
    for (pa, xa), (pb, xb) in zip(O.tree_leaves_with_path(ga), O.tree_leaves_with_path(gb)):
        assert pa == pb
        np.testing.assert_allclose(xa, xb.numpy(), rtol=1e-9, atol=1e-12, err_msg=pa)


def test_numpy_matches_torch_update_fp32_multi_step():
    cfg, params, batch, noise = make_case(dtype=np.float32)
    a = O.OracleFQL(params, cfg, 7, 3, np.float32)
    b = TorchFQL(params, cfg, torch.float32)
    for s in range(5):
        noise = O.make_noise(16, 3, 50 + s)
        _, ia = a.update(batch, noise)
        _, ib = b.update(batch, noise)
        for k in O.INFO_KEYS:
            assert abs(ia[k] - ib[k]) <= 2e-5 * max(1.0, abs(ib[k])), (s, k, ia[k], ib[k])
    for (pa, xa), (_, xb) in zip(O.tree_leaves_with_path(a.params), O.tree_leaves_with_path(b.params)):
        np.testing.assert_allclose(xa, xb.numpy(), rtol=1e-4, atol=2e-6, err_msg=pa)


def test_fd_on_isolated_terms():
    """FD check of each differentiated path with the stored-param uses frozen, which is the
    function jax.grad actually sees (utils/flax_utils.py:90-118,137)."""
    cfg, params, batch, noise = make_case()
    a = O.OracleFQL(params, cfg, 7, 3, np.float64)
    _, _, g = a.grads(batch, noise)
    gl = dict(O.tree_leaves_with_path(g))
    frozen = TorchFQL(params, cfg, torch.float64)

    def loss_with(path, idx, delta):
        gp = O.tree_map(lambda t: t.clone(), frozen.params)
        node = gp
        keys = path.split('/')
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]][idx] += delta
        l, _ = frozen.total_loss(batch, noise, grad_params=gp)
        return float(l)

    rng = np.random.default_rng(5)
    h = 1e-6
    for path, leaf in O.tree_leaves_with_path(a.params):
        for _ in range(3):
            idx = tuple(int(rng.integers(0, s)) for s in leaf.shape)
            fd = (loss_with(path, idx, h) - loss_with(path, idx, -h)) / (2 * h)
            assert abs(fd - gl[path][idx]) < 1e-6 * max(1.0, abs(fd)), (path, idx, fd, gl[path][idx])


def test_kat_gelu_is_tanh_approximation():
    # F2: flax nn.gelu == tanh form, not erf: gelu_tanh(1) = 0.841191990608 (erf form: 0.841344746)
    assert abs(float(O.gelu_tanh(np.float64(1.0))) - 0.8411919906082768) < 1e-15
    x = np.linspace(-4, 4, 101)
    np.testing.assert_allclose(O.gelu_tanh(x), torch.nn.functional.gelu(torch.tensor(x), approximate='tanh').numpy(), atol=1e-14)


def test_kat_layer_order_dense_gelu_ln_and_no_ln_after_last():
    # F3
    net = {'Dense_0': {'kernel': np.array([[2.0, -1.0]]), 'bias': np.array([0.5, 0.25])},
           'LayerNorm_0': {'scale': np.array([1.5, 0.5]), 'bias': np.array([0.1, -0.1])},
           'Dense_1': {'kernel': np.array([[1.0], [3.0]]), 'bias': np.array([0.0])}}
    x = np.array([[1.0]])
    g = O.gelu_tanh(np.array([[2.5, -0.75]]))
    mean = g.mean(); var = (g * g).mean() - mean ** 2
    ln = (g - mean) / np.sqrt(var + 1e-6) * np.array([1.5, 0.5]) + np.array([0.1, -0.1])
    expect = ln @ np.array([[1.0], [3.0]])
    np.testing.assert_allclose(O.mlp_forward(net, x), expect, rtol=1e-14)


def test_kat_layernorm_eps_and_fast_variance_clamp():
    x = np.full((1, 8), 3.0, dtype=np.float32)  # E[x^2]-E[x]^2 may round negative -> clamped to 0
    y, xhat, rstd = O.layer_norm(x, np.ones(8, np.float32), np.zeros(8, np.float32))
    assert np.all(np.isfinite(y)) and abs(float(rstd[0, 0]) - 1000.0) < 1e-1  # 1/sqrt(1e-6)


def test_kat_polyak_uses_pre_step_critic_and_adam_bias_correction():
    # F4 + optax.adam count semantics
    cfg, params, batch, noise = make_case(dtype=np.float64)
    a = O.OracleFQL(params, cfg, 7, 3, np.float64)
    old = copy.deepcopy(a.params)
    _, _, g = a.grads(batch, noise)
    a.apply_gradients(g)
    k_old = old['modules_critic']['value_net']['Dense_1']['kernel']
    t_old = old['modules_target_critic']['value_net']['Dense_1']['kernel']
    np.testing.assert_allclose(a.params['modules_target_critic']['value_net']['Dense_1']['kernel'],
                               0.005 * k_old + 0.995 * t_old, rtol=1e-15)
    # first Adam step: m_hat = g, v_hat = g^2  =>  delta = -lr * g/(|g|+eps)
    gk = g['modules_critic']['value_net']['Dense_1']['kernel']
    np.testing.assert_allclose(a.params['modules_critic']['value_net']['Dense_1']['kernel'],
                               k_old - 3e-4 * gk / (np.abs(gk) + 1e-8), rtol=1e-9, atol=1e-15)
    assert a.step == 2 and a.count == 1


def test_kat_grad_stats_definition_and_target_leaves_zero():
    # F5: grad/norm is the SUM of per-leaf L2 norms; target leaves have zero grads and take part
    cfg, params, batch, noise = make_case()
    a = O.OracleFQL(params, cfg, 7, 3, np.float64)
    _, _, g = a.grads(batch, noise)
    leaves = O.tree_leaves_with_path(g)
    assert len(leaves) == 56
    for p, x in leaves:
        if 'target' in p:
            assert not x.any()
    st = O.OracleFQL.grad_stats(g)
    assert abs(st['grad/norm'] - sum(np.linalg.norm(x.ravel()) for _, x in leaves)) < 1e-12
    glob = np.sqrt(sum(np.sum(x * x) for _, x in leaves))
    assert st['grad/norm'] > glob  # L1-of-norms, not a global L2
    assert st['grad/max'] >= 0 >= st['grad/min']


def test_kat_clip_gradient_mask_and_q_agg():
    cfg, params, batch, noise = make_case(B=8)
    # blow up the one-step actor's last bias so every action saturates -> Q term passes no grad
    params['modules_actor_onestep_flow']['mlp']['Dense_4']['bias'][:] = 50.0
    a = O.OracleFQL(params, cfg, 7, 3, np.float64)
    b = TorchFQL(params, cfg, torch.float64)
    _, _, ga = a.grads(batch, noise)
    _, _, gb = b.grads(batch, noise)
    for (p, xa), (_, xb) in zip(O.tree_leaves_with_path(ga), O.tree_leaves_with_path(gb)):
        np.testing.assert_allclose(xa, xb.numpy(), rtol=1e-9, atol=1e-12, err_msg=p)
    acts = a.sample_actions(batch['observations'], noise['eps2'])
    assert np.all(acts == 1.0)


def test_sample_actions_shapes_and_flow_actions():
    cfg, params, batch, noise = make_case()
    a = O.OracleFQL(params, cfg, 7, 3, np.float64)
    one = a.sample_actions(batch['observations'][0], noise['eps2'][0])  # 1-D obs (main.py:225)
    assert one.shape == (3,)
    fa = a.compute_flow_actions(batch['observations'], noise['z'])
    assert fa.shape == (16, 3) and np.all(np.abs(fa) <= 1)
