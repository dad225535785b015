"""CPU: the numpy encoder oracle against an independent torch-autograd restatement (conv2d / max_pool2d / gelu)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import encoder_oracle as E


def torch_forward(p, x_u8):
    x = torch.as_tensor(np.asarray(x_u8), dtype=torch.float64) / 255.0
    x = x.permute(0, 3, 1, 2)
    ns = sum(1 for k in p if k.startswith('stack_blocks_'))

    def conv(x, c):
        return F.conv2d(x, c['kernel'].permute(3, 2, 0, 1), c['bias'], padding=1)

    for s in range(ns):
        st = p[f'stack_blocks_{s}']
        x = conv(x, st['Conv_0'])
        x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), value=float('-inf')), 3, 2)
        for b in range((len(st) - 1) // 2):
            inp = x
            x = conv(F.relu(x), st[f'Conv_{1 + 2 * b}'])
            x = conv(F.relu(x), st[f'Conv_{2 + 2 * b}'])
            x = x + inp
    x = F.relu(x).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    for i in range(len(p['MLP_0'])):
        d = p['MLP_0'][f'Dense_{i}']
        x = F.gelu(x @ d['kernel'] + d['bias'], approximate='tanh')
    return x


def to_torch(tree):
    if isinstance(tree, dict):
        return {k: to_torch(v) for k, v in tree.items()}
    return torch.tensor(np.asarray(tree, np.float64), requires_grad=True)


def flat(tree, prefix=''):
    out = {}
    for k in sorted(tree):
        v = tree[k]
        if isinstance(v, dict):
            out.update(flat(v, f'{prefix}{k}/'))
        else:
            out[prefix + k] = v
    return out


@pytest.mark.parametrize('name,hw,c', [('impala_small', 16, 9), ('impala_debug', 8, 3), ('impala', 16, 3)])
def test_encoder_forward_backward_match_torch(name, hw, c):
    rng = np.random.default_rng(0)
    p = E.init_encoder_params(rng, (hw, hw, c), name, np.float64)
    for path, leaf in flat(p).items():          # non-zero biases so their gradients are exercised
        if path.endswith('bias'):
            leaf[...] = rng.normal(size=leaf.shape) * 0.1
    x = rng.integers(0, 256, size=(3, hw, hw, c), dtype=np.uint8)
    out, cache = E.impala_forward(p, x, keep=True, dtype=np.float64)
    tp = to_torch(p)
    tout = torch_forward(tp, x)
    np.testing.assert_allclose(out, tout.detach().numpy(), rtol=1e-10, atol=1e-12)
    dout = rng.normal(size=out.shape)
    g = E.impala_backward(p, cache, dout)
    (tout * torch.as_tensor(dout)).sum().backward()
    fg, ft = flat(g), flat(tp)
    assert set(fg) == set(ft)
    for path in fg:
        np.testing.assert_allclose(fg[path], ft[path].grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=path)


def test_leaf_names_and_shapes_impala_small():
    p = E.init_encoder_params(np.random.default_rng(1), (64, 64, 9), 'impala_small')
    f = flat(p)
    assert f['stack_blocks_0/Conv_0/kernel'].shape == (3, 3, 9, 16)
    assert f['stack_blocks_1/Conv_0/kernel'].shape == (3, 3, 16, 32)
    assert f['stack_blocks_2/Conv_2/kernel'].shape == (3, 3, 32, 32)
    assert f['MLP_0/Dense_0/kernel'].shape == (2048, 512)      # 8 * 8 * 32
    assert len(f) == 2 * 9 + 2


def test_max_pool_same_padding_sits_at_the_end():
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    y, arg = E.max_pool(x)
    np.testing.assert_array_equal(y[0, :, :, 0], [[10, 11], [14, 15]])
    np.testing.assert_array_equal(arg[0, :, :, 0], [[8, 7], [5, 4]])   # last window row/col only see 2 valid rows/cols


def test_stack_frames_clamps_to_episode_start():
    n = 10
    frames = np.arange(n, dtype=np.uint8).reshape(n, 1, 1, 1)
    nxt = frames + 100
    terminals = np.zeros(n); terminals[4] = 1; terminals[9] = 1          # episodes [0..4], [5..9]
    obs, nobs = E.stack_frames(frames, nxt, terminals, np.array([0, 1, 5, 6, 9]), 3)
    np.testing.assert_array_equal(obs[:, 0, 0, :], [[0, 0, 0], [0, 0, 1], [5, 5, 5], [5, 5, 6], [7, 8, 9]])
    np.testing.assert_array_equal(nobs[:, 0, 0, :], [[0, 0, 100], [0, 1, 101], [5, 5, 105], [5, 6, 106], [8, 9, 109]])


def test_random_crop_matches_edge_pad_slice():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(2, 8, 8, 3), dtype=np.uint8)
    out = E.random_crop_batch(img, np.array([[3, 3], [0, 6]]))
    np.testing.assert_array_equal(out[0], img[0])                          # offset == padding: identity
    np.testing.assert_array_equal(out[1, 3:, :5], img[1, :5, 3:])          # shifted down 3, left 3
    np.testing.assert_array_equal(out[1, 0, :5], img[1, 0, 3:])            # top rows replicate the edge


def make_visual_problem(B=4, hw=16, c=3, ad=4, enc='impala_debug', hidden=(32, 32), seed=0, dtype=np.float64):
    from oracle import fql_oracle as O
    cfg = O.get_config()
    cfg.update(encoder=enc, actor_hidden_dims=hidden, value_hidden_dims=hidden, alpha=3.0)
    rng = np.random.default_rng(seed)
    batch = {
        'observations': rng.integers(0, 256, size=(B, hw, hw, c), dtype=np.uint8),
        'next_observations': rng.integers(0, 256, size=(B, hw, hw, c), dtype=np.uint8),
        'actions': rng.uniform(-1, 1, size=(B, ad)).astype(np.float32),
        'rewards': -np.ones(B, np.float32), 'masks': np.ones(B, np.float32),
    }
    agent = O.OracleFQL.create(seed + 1, (hw, hw, c), ad, cfg, dtype)
    return O, cfg, agent, batch, O.make_noise(B, ad, seed + 2)


def test_visual_total_loss_gradients_match_finite_differences():
    """jax.grad over the whole tree (utils/flax_utils.py:137) incl. the three encoders, against central differences."""
    O, cfg, agent, batch, noise = make_visual_problem()
    loss, info, grads = agent.grads(batch, noise)
    agent.frozen = O.tree_map(lambda a: a.copy(), agent.params)   # stored-param uses stay put while a leaf is perturbed
    leaves = dict(O.tree_leaves_with_path(agent.params))
    gl = dict(O.tree_leaves_with_path(grads))
    assert any('/encoder/' in k for k in gl)
    assert all(np.all(v == 0) for k, v in gl.items() if k.startswith('modules_target_critic'))
    rng = np.random.default_rng(5)
    paths = [k for k in leaves if not k.startswith('modules_target_critic')]
    checked, kinks = 0, 0
    for path in paths:
        leaf = leaves[path]
        for _ in range(2):
            idx = tuple(rng.integers(0, n) for n in leaf.shape)
            old = leaf[idx]
            h = 1e-6
            leaf[idx] = old + h; lp, _ = agent.total_loss(batch, noise)
            leaf[idx] = old - h; lm, _ = agent.total_loss(batch, noise)
            leaf[idx] = old
            fd = (lp - lm) / (2 * h)
            err = abs(fd - gl[path][idx])
            assert err <= 1e-6 + 2e-2 * abs(fd), (path, idx, fd, gl[path][idx])
            kinks += err > 1e-7 + 1e-4 * abs(fd)        # a ReLU / max-pool switch inside +-h (piecewise-linear nets)
            checked += 1
    assert checked >= 100 and kinks <= checked // 20


def test_visual_update_runs_and_moves_every_trainable_leaf():
    O, cfg, agent, batch, noise = make_visual_problem(dtype=np.float32)
    before = {k: v.copy() for k, v in O.tree_leaves_with_path(agent.params)}
    loss, info = agent.update(batch, noise)
    assert set(info) == set(O.INFO_KEYS) and all(np.isfinite(v) for v in info.values())
    after = dict(O.tree_leaves_with_path(agent.params))
    for k in before:
        if k.endswith('kernel'):
            assert not np.array_equal(before[k], after[k]), k


def test_visual_numpy_oracle_equals_torch_autograd_restatement():
    """The hand-derived visual backward (fql_oracle + encoder_oracle) against torch.autograd over F.conv2d / F.max_pool2d."""
    from oracle.fql_oracle_torch import TorchFQL
    O, cfg, agent, batch, noise = make_visual_problem(B=4, hw=16, c=3, enc='impala_debug')
    tref = TorchFQL(agent.params, cfg, torch.float64)
    loss, info, grads = agent.grads(batch, noise)
    tl, tinfo, tg = tref.grads(batch, noise)
    assert abs(float(tl) - loss) <= 1e-10 * abs(loss)
    for k, v in tinfo.items():
        assert abs(float(v.detach()) - info[k]) <= 1e-9 * (1 + abs(info[k])), k
    gl = dict(O.tree_leaves_with_path(grads))
    for path, g in O.tree_leaves_with_path(tg):
        np.testing.assert_allclose(g.numpy(), gl[path], rtol=1e-8, atol=1e-11, err_msg=path)
    # three updates stay together (Adam, Polyak incl. the encoder leaves)
    for s in range(3):
        nz = O.make_noise(4, 4, 30 + s)
        _, ia = agent.update(batch, nz)
        _, ib = tref.update(batch, nz)
        for k in ia:
            assert abs(ia[k] - ib[k]) <= 1e-8 * (1 + abs(ia[k])), k
