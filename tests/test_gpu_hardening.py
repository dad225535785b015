"""GPU parity at the sizes that are benchmarked, and checks of the production (device-side) RNG / gather / init path.

* per-leaf gradients and post-step parameters against the fp64 oracle at BASELINE.json configs[1] (obs 29, act 8, B 256,
  hidden 512x4) and configs[2] (obs 40, act 4, B 1024, alpha 300): the shapes at which the Euler-chain kernel, the 32 x 32 /
  32 x 64 / 64 x 64 side tiles and the three-lane program actually run (the small cases never reach them);
* Glorot-uniform bounds / zero biases / LayerNorm = (1, 0) / target := critic of `FQLAgent.create`
  (utils/networks.py:9-11, agents/fql.py:241-242);
* the device Philox streams behind update_from_dataset: moments and Kolmogorov-Smirnov distance of the five noise tensors
  (agents/fql.py:52-54,62-63,144-150), index draw inside the shard and uniform (utils/datasets.py:64-66), per-step and
  per-seed independence; for image datasets the crop offsets in [0, 2 pad] and ONE augmentation coin per batch
  (utils/datasets.py:88-92,102-112).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import assert_info_close, leaf_dict, make_problem, randomize_params

pytestmark = pytest.mark.gpu


def _agent(cfg, batch, seed=0):
    import fql_amd
    return fql_amd.FQLAgent.create(seed, batch['observations'][:1], batch['actions'][:1], cfg)


from tests.util import assert_step_matches as _assert_step_matches  # noqa: E402  (shared with the data-parallel program tests)


@pytest.mark.parametrize('od,ad,B,alpha', [(29, 8, 256, 10.0), (40, 4, 1024, 300.0)], ids=['configs1', 'configs2'])
def test_full_size_per_leaf_gradients_and_post_step_params(od, ad, B, alpha):
    cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=41, alpha=alpha)
    agent = _agent(cfg, batch)
    params = randomize_params(agent.get_params(), seed=9, scale=0.05)
    agent.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    worst = _assert_step_matches(agent, ref, cfg, batch, noise)
    assert worst[0] <= 2e-5, worst   # max over leaves of max|g_gpu - g_ref| / max|g_ref|


def test_create_initialises_like_the_reference():
    """utils/networks.py:9-11 (variance_scaling(1, fan_avg, uniform) = Glorot uniform), flax Dense bias 0, LayerNorm scale 1 / bias 0,
    agents/fql.py:241-242 target_critic := critic; different seeds give different kernels."""
    cfg, ds, batch, noise = make_problem(29, 8, 256, (512, 512, 512, 512), seed=1)
    a0, a1 = _agent(cfg, batch, seed=0), _agent(cfg, batch, seed=1)
    p0, p1 = a0.get_params(), a1.get_params()
    n_kernel = 0
    for path, w in O.tree_leaves_with_path(p0):
        if path.endswith('/kernel'):
            fan_in, fan_out = w.shape[-2], w.shape[-1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            assert np.abs(w).max() <= lim * (1 + 1e-6), path
            if w.size >= 4096:
                assert np.abs(w).max() >= 0.98 * lim, path                       # the bound is attained, not a smaller box
                assert abs(w.var() / (lim * lim / 3.0) - 1.0) < 0.05, (path, w.var())  # U(-lim, lim): variance lim^2 / 3
                assert abs(w.mean()) < 4 * lim / np.sqrt(3 * w.size), path
            if w.ndim == 3:   # ensemble members are drawn independently
                assert np.abs(w[0] - w[1]).max() > 0.1 * lim, path
            n_kernel += 1
        elif 'LayerNorm' in path and path.endswith('/scale'):
            np.testing.assert_array_equal(w, 1.0, err_msg=path)
        else:   # Dense bias, LayerNorm bias
            np.testing.assert_array_equal(w, 0.0, err_msg=path)
    assert n_kernel == 20
    for (path, a), (_, b) in zip(O.tree_leaves_with_path(p0['modules_critic']), O.tree_leaves_with_path(p0['modules_target_critic'])):
        np.testing.assert_array_equal(a, b, err_msg=path)
    k = 'modules_actor_bc_flow'
    assert np.abs(leaf_dict(p0[k])['mlp/Dense_1/kernel'] - leaf_dict(p1[k])['mlp/Dense_1/kernel']).max() > 0.01
    opt = a0.get_opt_state()
    assert opt['count'] == 0 and opt['step'] == 1           # utils/flax_utils.py:81: TrainState.step starts at 1
    for path, m in O.tree_leaves_with_path(opt['mu']):
        assert not m.any(), path


def _workspace(agent, which, dtype=np.float32):
    from fql_amd import _cabi
    lib = _cabi.load()
    f = lib.fql_debug_workspace
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    dims = (C.c_int * 2)()
    buf = np.empty(1 << 22, dtype=dtype)
    rc = f(agent._h, which, buf.ctypes.data, buf.nbytes, dims)
    assert rc == 0, rc
    return buf[:dims[0] * dims[1]].reshape(dims[0], dims[1]).copy()


def _draws(agent, B, od, ad):
    """The five noise tensors and the gathered rows of the LAST update, recovered from the network inputs the prep kernel built."""
    x_os, x_bc, vel, act, x_c1 = (_workspace(agent, w) for w in (0, 1, 2, 3, 4))
    eps1, z, eps2 = (x_os[i * B:(i + 1) * B, od:od + ad] for i in range(3))
    x0 = act[:, :ad] - vel[:, :ad]                   # vel = a - x0 (agents/fql.py:56)
    t = x_bc[:, od + ad]
    return dict(eps1=eps1, x0=x0, t=t, z=z, eps2=eps2), x_c1[:, 0].copy(), x_c1


def test_device_rng_streams_are_standard_normal_uniform_and_independent():
    from scipy import stats
    import fql_amd
    od, ad, B, N = 5, 8, 4096, 50_000
    cfg = fql_amd.get_config()
    cfg.update(actor_hidden_dims=(32, 32, 32), value_hidden_dims=(32, 32), alpha=10.0, batch_size=B)
    ds = O.make_synthetic_dataset(N, od, ad, seed=0)
    ds['observations'][:, 0] = np.arange(N, dtype=np.float32)          # row id in the first observation column
    lo, hi = 10_000, 30_000
    seen = {}
    for seed in (0, 1):
        agent = fql_amd.FQLAgent.create(seed, ds['observations'][:1], ds['actions'][:1], cfg)
        agent.upload_dataset(ds)
        per_step = []
        for step in range(3):
            agent.update_from_dataset(B, shard=(lo, hi))
            noise, rows, x_c1 = _draws(agent, B, od, ad)
            per_step.append((noise, rows))
            idx = rows.astype(np.int64)
            assert idx.min() >= lo and idx.max() < hi                   # utils/datasets.py:66 inside this rank's shard
            np.testing.assert_array_equal(x_c1[:, 1:od], ds['observations'][idx, 1:od])   # whole rows gathered, not just ids
            counts = np.bincount((idx - lo) * 16 // (hi - lo), minlength=16)
            assert stats.chisquare(counts).pvalue > 1e-4, counts      # uniform over the shard
            assert len(np.unique(idx)) > 0.85 * B                       # i.i.d. draws with replacement: ~9 % repeats at 4096 of 20000
            for k in ('eps1', 'x0', 'z', 'eps2'):
                v = noise[k].astype(np.float64).ravel()                 # 32768 samples
                assert abs(v.mean()) < 4.5 / np.sqrt(v.size), (k, v.mean())
                assert abs(v.var() - 1.0) < 0.03, (k, v.var())
                assert abs(stats.skew(v)) < 0.06 and abs(stats.kurtosis(v)) < 0.12, k
                assert stats.kstest(v, 'norm').statistic < 0.012, k     # 1e-4 critical value at n = 32768 is ~0.012
                cols = noise[k].astype(np.float64)
                cc = np.corrcoef(cols.T) - np.eye(ad)
                assert np.abs(cc).max() < 0.08, k                       # action components uncorrelated (Box-Muller pairs included)
            tt = noise['t'].astype(np.float64)
            assert tt.min() >= 0.0 and tt.max() < 1.0
            assert stats.kstest(tt, 'uniform').statistic < 0.035
            names = ['eps1', 'x0', 'z', 'eps2']
            for i in range(4):
                for j in range(i + 1, 4):
                    r = np.corrcoef(noise[names[i]].ravel(), noise[names[j]].ravel())[0, 1]
                    assert abs(r) < 0.03, (names[i], names[j], r)      # the five tensors come from different streams
        for a in range(3):
            for b in range(a + 1, 3):                                   # a new stream every update
                assert not np.array_equal(per_step[a][0]['z'], per_step[b][0]['z'])
                assert abs(np.corrcoef(per_step[a][0]['z'].ravel(), per_step[b][0]['z'].ravel())[0, 1]) < 0.03
                assert not np.array_equal(per_step[a][1], per_step[b][1])
        seen[seed] = per_step[0]
        agent.close()
    assert not np.array_equal(seen[0][0]['z'], seen[1][0]['z'])        # keyed by the agent seed
    assert not np.array_equal(seen[0][1], seen[1][1])


def test_frames_gather_crop_offsets_and_single_coin():
    """utils/datasets.py:88-92: `if np.random.rand() < p_aug: augment(...)` is ONE coin per batch; random_crop offsets are per sample in
    [0, 2 * padding] (padding 3), the same for observations and next_observations (utils/datasets.py:102-112)."""
    import fql_amd
    n, ad, B = 2048, 5, 64
    rng = np.random.default_rng(0)
    term = (rng.random(n) < 1.0 / 100).astype(np.float32); term[-1] = 1
    ds = {'observations': rng.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8),
          'next_observations': rng.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8),
          'actions': rng.uniform(-1, 1, size=(n, ad)).astype(np.float32), 'rewards': -np.ones(n, np.float32), 'masks': 1 - term,
          'terminals': term}
    cfg = fql_amd.get_config()
    cfg.update(actor_hidden_dims=(64, 64, 64), value_hidden_dims=(64, 64), alpha=300.0, batch_size=B, encoder='impala_small')
    agent = fql_amd.FQLAgent.create(0, np.zeros((1, 32, 32, 9), np.uint8), ds['actions'][:1], cfg)
    agent.upload_dataset(ds, frame_stack=3, p_aug=0.5)
    coins, offs = [], []
    for _ in range(40):
        agent.update_from_dataset(B)
        crop = _workspace(agent, 8, np.int32)
        idx = _workspace(agent, 7, np.int64)[:B, 0]
        assert idx.min() >= 0 and idx.max() < n
        assert crop.min() >= 0 and crop.max() <= 6
        identity = np.all(crop == 3)                 # crop_from == padding is the un-augmented slice
        coins.append(not identity)
        if not identity:
            offs.append(crop.copy())
            assert len(np.unique(crop[:, 0])) >= 5 and len(np.unique(crop[:, 1])) >= 5   # per-sample offsets, not one per batch
    assert 8 <= sum(coins) <= 32, sum(coins)         # p_aug = 0.5 over 40 batches (binomial: < 1e-4 outside)
    o = np.concatenate(offs)
    counts = np.bincount(o.ravel(), minlength=7)
    from scipy import stats
    assert stats.chisquare(counts).pvalue > 1e-4, counts
