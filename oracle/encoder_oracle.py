"""CPU oracle of the reference's IMPALA encoder (visual path).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product path (fql_amd/) never imports it.

PARITY UNPINNED: the reference (utils/encoders.py) is flax code and flax / jax are absent from this image, so
nothing here was checked against the reference running.  It restates the published semantics of the library calls
the reference makes, each cited below:

  * ``nn.Conv(features, (3, 3), strides=1, padding='SAME', kernel_init=xavier_uniform)`` (utils/encoders.py:19-25,
    39-45, 48-54): cross-correlation, NHWC, kernel leaf [3, 3, Cin, Cout] (HWIO), zero padding 1 on every side,
    bias leaf [Cout] (zeros at init);  y[n,h,w,o] = b[o] + sum_{i,j,c} x[n,h+i-1,w+j-1,c] K[i,j,c,o].
  * ``nn.max_pool(x, (3, 3), strides=(2, 2), padding='SAME')`` (utils/encoders.py:27-33): -inf padding; for an even
    extent H the output extent is H/2 and the single padded row/column sits at the END (total padding 1, low side
    0), so window o covers input rows 2o .. 2o+2.  Gradient: to the window's maximum (first in row-major window
    order on ties, as XLA's select-and-scatter with a >= select).
  * ResnetStack (utils/encoders.py:10-58): conv -> max_pool -> num_blocks x [relu, conv, relu, conv, + block input].
  * ImpalaEncoder (utils/encoders.py:61-100): x / 255; stacks; relu; (LayerNorm if layer_norm); flatten in (h, w, c)
    order; MLP(mlp_hidden_dims, activate_final=True, layer_norm=layer_norm) = Dense + GELU(tanh) (+ LayerNorm).
    impala_small = num_blocks 1, stack_sizes (16, 32, 32), mlp (512,), no LayerNorm, no dropout (utils/encoders.py:106).
  * frame stacking / random crop (utils/datasets.py:17-33, 73-112): see ``stack_frames`` / ``random_crop_batch``.

Parameter names follow flax's auto-naming: ``stack_blocks_{s}/Conv_{j}/{kernel,bias}`` (setup list attribute +
compact submodules) and ``MLP_0/Dense_0/{kernel,bias}``.
"""
from __future__ import annotations

import math

import numpy as np

ENCODERS = {
    # name: (stack_sizes, num_blocks, mlp_hidden_dims)      utils/encoders.py:103-108
    'impala': ((16, 32, 32), 2, (512,)),
    'impala_debug': ((4, 4), 1, (512,)),
    'impala_small': ((16, 32, 32), 1, (512,)),
    'impala_large': ((64, 128, 128), 2, (1024,)),
}


def gelu_tanh(x):
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + np.tanh(c * (x + 0.044715 * x ** 3)))


def gelu_tanh_grad(x):
    c = math.sqrt(2.0 / math.pi)
    u = c * (x + 0.044715 * x ** 3)
    th = np.tanh(u)
    return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * c * (1.0 + 3 * 0.044715 * x * x)


# ----------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------
def _patches(x):
    """[N,H,W,C] -> [N,H,W,3,3,C] zero-padded 3x3 neighbourhoods (copy-free view of the padded array)."""
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    return np.lib.stride_tricks.sliding_window_view(xp, (3, 3), axis=(1, 2)).transpose(0, 1, 2, 4, 5, 3)


def conv3x3(x, kernel, bias):
    return np.einsum('nhwijc,ijco->nhwo', _patches(x), kernel, optimize=True) + bias


def conv3x3_bwd(x, kernel, dy):
    """Returns (dx, dkernel, dbias)."""
    dk = np.einsum('nhwijc,nhwo->ijco', _patches(x), dy, optimize=True)
    db = dy.sum(axis=(0, 1, 2))
    # dx[n,h,w,c] = sum_{i,j,o} dy[n,h-i+1,w-j+1,o] K[i,j,c,o]  = conv of dy with the flipped, transposed kernel
    kf = kernel[::-1, ::-1].transpose(0, 1, 3, 2)
    dx = np.einsum('nhwijo,ijoc->nhwc', _patches(dy), kf, optimize=True)
    return dx, dk, db


def max_pool(x):
    """3x3 / stride 2 / SAME on even extents.  Returns (y, arg) with arg = winning offset 3*i + j per output."""
    n, h, w, c = x.shape
    assert h % 2 == 0 and w % 2 == 0
    xp = np.pad(x, ((0, 0), (0, 1), (0, 1), (0, 0)), constant_values=-np.inf)
    win = np.lib.stride_tricks.sliding_window_view(xp, (3, 3), axis=(1, 2))[:, ::2, ::2]  # [N,H/2,W/2,C,3,3]
    flat = win.reshape(n, h // 2, w // 2, c, 9)
    arg = flat.argmax(axis=-1)  # first maximum in row-major window order
    return np.take_along_axis(flat, arg[..., None], axis=-1)[..., 0], arg.astype(np.uint8)


def max_pool_bwd(arg, dy, in_hw):
    n, ho, wo, c = dy.shape
    h, w = in_hw
    dx = np.zeros((n, h + 1, w + 1, c), dy.dtype)
    oh, ow = np.meshgrid(np.arange(ho), np.arange(wo), indexing='ij')
    for k in range(9):
        i, j = divmod(k, 3)
        m = (arg == k)
        # windows do not overlap at a fixed offset k, so a direct add per offset is a correct scatter
        dx[:, 2 * oh + i, 2 * ow + j, :] += dy * m
    return dx[:, :h, :w, :]


# ----------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------
def init_encoder_params(rng: np.random.Generator, in_hwc, name='impala_small', dtype=np.float32) -> dict:
    """xavier_uniform conv kernels (fan_in = 9 Cin, fan_out = 9 Cout), zero biases; Dense as utils/networks.py:9-11."""
    stack_sizes, num_blocks, mlp_dims = ENCODERS[name]
    h, w, c = in_hwc
    p = {}
    cin = c
    for s, feat in enumerate(stack_sizes):
        st = {}
        for j in range(1 + 2 * num_blocks):
            lim = math.sqrt(6.0 / (9 * cin + 9 * feat))
            st[f'Conv_{j}'] = {'kernel': rng.uniform(-lim, lim, size=(3, 3, cin, feat)).astype(dtype),
                               'bias': np.zeros((feat,), dtype)}
            cin = feat
        p[f'stack_blocks_{s}'] = st
        h, w = h // 2, w // 2
    d_in = h * w * cin
    mlp = {}
    for i, d_out in enumerate(mlp_dims):
        lim = math.sqrt(6.0 / (d_in + d_out))
        mlp[f'Dense_{i}'] = {'kernel': rng.uniform(-lim, lim, size=(d_in, d_out)).astype(dtype),
                             'bias': np.zeros((d_out,), dtype)}
        d_in = d_out
    p['MLP_0'] = mlp
    return p


def encoder_out_dim(name='impala_small') -> int:
    return ENCODERS[name][2][-1]


# ----------------------------------------------------------------------------------------
# forward / backward
# ----------------------------------------------------------------------------------------
def impala_forward(p: dict, x_u8, keep=False, dtype=np.float32):
    """utils/encoders.py:83-100.  x_u8: [N,H,W,C] uint8 (or float already in 0..255)."""
    x = np.asarray(x_u8).astype(dtype) / dtype(255.0) if np.dtype(dtype) != np.float64 else np.asarray(x_u8).astype(np.float64) / 255.0
    cache = {'stacks': []}
    ns = sum(1 for k in p if k.startswith('stack_blocks_'))
    for s in range(ns):
        st = p[f'stack_blocks_{s}']
        nb = (len(st) - 1) // 2
        sc = {'x0': x}
        y = conv3x3(x, st['Conv_0']['kernel'], st['Conv_0']['bias'])
        sc['pre_pool_hw'] = y.shape[1:3]
        y, arg = max_pool(y)
        sc['arg'] = arg
        sc['blocks'] = []
        for b in range(nb):
            inp = y
            r1 = np.maximum(inp, 0)
            c1 = conv3x3(r1, st[f'Conv_{1 + 2 * b}']['kernel'], st[f'Conv_{1 + 2 * b}']['bias'])
            r2 = np.maximum(c1, 0)
            c2 = conv3x3(r2, st[f'Conv_{2 + 2 * b}']['kernel'], st[f'Conv_{2 + 2 * b}']['bias'])
            y = c2 + inp
            sc['blocks'].append((inp, c1))
        cache['stacks'].append(sc)
        x = y
    cache['final'] = x
    f = np.maximum(x, 0).reshape(x.shape[0], -1)
    cache['flat'] = f
    mlp = p['MLP_0']
    zs = []
    h = f
    for i in range(len(mlp)):
        z = h @ mlp[f'Dense_{i}']['kernel'] + mlp[f'Dense_{i}']['bias']
        zs.append((h, z))
        h = gelu_tanh(z)
    cache['mlp'] = zs
    return (h, cache) if keep else h


def impala_backward(p: dict, cache: dict, dout):
    """Gradient of ``impala_forward`` w.r.t. its parameters (input images carry no gradient)."""
    g = {}
    mlp = p['MLP_0']
    gm = {}
    dh = dout
    for i in reversed(range(len(mlp))):
        h, z = cache['mlp'][i]
        dz = dh * gelu_tanh_grad(z)
        gm[f'Dense_{i}'] = {'kernel': h.T @ dz, 'bias': dz.sum(axis=0)}
        dh = dz @ mlp[f'Dense_{i}']['kernel'].T
    g['MLP_0'] = gm
    x = cache['final']
    dy = dh.reshape(x.shape) * (x > 0)
    ns = len(cache['stacks'])
    for s in reversed(range(ns)):
        st = p[f'stack_blocks_{s}']
        sc = cache['stacks'][s]
        gs = {}
        for b in reversed(range(len(sc['blocks']))):
            inp, c1 = sc['blocks'][b]
            r2 = np.maximum(c1, 0)
            d_r2, dk, db = conv3x3_bwd(r2, st[f'Conv_{2 + 2 * b}']['kernel'], dy)
            gs[f'Conv_{2 + 2 * b}'] = {'kernel': dk, 'bias': db}
            d_c1 = d_r2 * (c1 > 0)
            r1 = np.maximum(inp, 0)
            d_r1, dk, db = conv3x3_bwd(r1, st[f'Conv_{1 + 2 * b}']['kernel'], d_c1)
            gs[f'Conv_{1 + 2 * b}'] = {'kernel': dk, 'bias': db}
            dy = dy + d_r1 * (inp > 0)
        d_conv = max_pool_bwd(sc['arg'], dy, sc['pre_pool_hw'])
        dx, dk, db = conv3x3_bwd(sc['x0'], st['Conv_0']['kernel'], d_conv)
        gs['Conv_0'] = {'kernel': dk, 'bias': db}
        g[f'stack_blocks_{s}'] = gs
        dy = dx
    return g


# ----------------------------------------------------------------------------------------
# dataset side (utils/datasets.py)
# ----------------------------------------------------------------------------------------
def stack_frames(frames, next_frames, terminals, idxs, frame_stack):
    """utils/datasets.py:73-88: obs = [ob[t-k+1..t]] clamped to the episode's first index, concatenated on the
    channel axis (oldest first); next_obs = [ob[t-k+2..t], next_ob[t]]."""
    idxs = np.asarray(idxs)
    terminal_locs = np.nonzero(terminals > 0)[0]
    initial_locs = np.concatenate([[0], terminal_locs[:-1] + 1])
    init = initial_locs[np.searchsorted(initial_locs, idxs, side='right') - 1]
    obs, nobs = [], []
    for i in reversed(range(frame_stack)):
        cur = np.maximum(idxs - i, init)
        obs.append(frames[cur])
        if i != frame_stack - 1:
            nobs.append(frames[cur])
    nobs.append(next_frames[idxs])
    return np.concatenate(obs, axis=-1), np.concatenate(nobs, axis=-1)


def random_crop_batch(imgs, crop_froms, padding=3):
    """utils/datasets.py:17-33, 102-112: edge-pad by ``padding`` and slice [H, W] from (crop_from_y, crop_from_x)."""
    n, h, w, c = imgs.shape
    pad = np.pad(imgs, ((0, 0), (padding, padding), (padding, padding), (0, 0)), mode='edge')
    out = np.empty_like(imgs)
    for b in range(n):
        y, x = int(crop_froms[b][0]), int(crop_froms[b][1])
        out[b] = pad[b, y:y + h, x:x + w]
    return out
