"""CPU, world_size 2 over gloo: the data-parallel contract of fql_amd/parallel.py.

A W-rank step with per-rank batch B must equal a 1-rank step on the concatenated W*B batch.  The engine
cannot run here (no GPU), so the oracle stands in as the per-rank gradient producer -- it is the checker for
the host-side logic being tested: row sharding, SUM all-reduce + 1/W scale, parameter broadcast helpers and
the metric reduction rules."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fql_amd.parallel import shard_range, tree_flatten, tree_unflatten
from oracle import fql_oracle as O


def test_shard_range_partitions_rows():
    for n, w in [(1_000_000, 8), (1000, 3), (17, 4), (8, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a < b
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_tree_flatten_roundtrip_matches_jax_key_order():
    params = O.init_params(0, 5, 2, dict(O.get_config(), actor_hidden_dims=(8, 8), value_hidden_dims=(8, 8)))
    items = tree_flatten(params)
    assert [p for p, _ in items] == [p for p, _ in O.tree_leaves_with_path(params)]
    back = tree_unflatten(items)
    for (p, a), (_, b) in zip(O.tree_leaves_with_path(params), O.tree_leaves_with_path(back)):
        assert a is b


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    od, ad, B = 7, 3, 8
    cfg = dict(O.get_config(), actor_hidden_dims=(16, 16), value_hidden_dims=(16, 16), alpha=10.0)
    params = O.init_params(3, od, ad, cfg, np.float64)
    ds = O.make_synthetic_dataset(64, od, ad, seed=0)
    lo, hi = shard_range(64, rank, world)
    idx = lo + np.random.default_rng(10 + rank).integers(0, hi - lo, size=B)     # local rows only
    noise = O.make_noise(B, ad, 20 + rank)
    ref = O.OracleFQL(params, cfg, od, ad, np.float64)
    _, info, g = ref.grads(O.sample_batch(ds, idx), noise)
    flat = torch.from_numpy(np.concatenate([v.reshape(-1) for _, v in tree_flatten(g)]))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)            # what DataParallelFQL does on the grad buffer
    flat /= world                                           # fql_set_grad_scale(1/W)
    gathered_idx = [torch.zeros(B, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered_idx, torch.from_numpy(idx))
    noises = {}
    for k, v in noise.items():
        parts = [torch.zeros_like(torch.from_numpy(v)) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(v))
        noises[k] = torch.cat(parts).numpy()
    if rank == 0:
        full_idx = torch.cat(gathered_idx).numpy()
        _, info_full, g_full = ref.grads(O.sample_batch(ds, full_idx), noises)
        want = np.concatenate([v.reshape(-1) for _, v in tree_flatten(g_full)])
        q.put((float(np.abs(flat.numpy() - want).max()), float(np.abs(want).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_mean_equals_concatenated_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err <= 1e-12 * max(1.0, scale), (err, scale)


# ---------------------------------------------------------------------------------------------------------
# DataParallelFQL ITSELF under gloo (world size 2): a CPU stand-in with the engine's data-parallel surface, its gradient
# buffer a CPU torch tensor, the oracle (fp64) doing the arithmetic.  What is exercised is the wrapper: state broadcast
# (params, Adam moments, count, step), the rank mixed into the RNG stream, the SUM all-reduce + 1/W scale placement
# between update_begin and update_end, shard bookkeeping, the normalize_q_loss refusal, metric reduction.
# ---------------------------------------------------------------------------------------------------------
class _CpuAgent:
    def __init__(self, seed, od, ad, cfg, ds):
        self.config = dict(cfg)
        self.od, self.ad = od, ad
        self.ref = O.OracleFQL(O.init_params(seed, od, ad, cfg, np.float64), cfg, od, ad, np.float64)
        self.ds, self.rows = ds, len(ds['observations'])
        self.layout = [(p, v.shape, v.size) for p, v in tree_flatten(self.ref.params) if 'target' not in p]
        self._g = torch.zeros(sum(n for _, _, n in self.layout), dtype=torch.float64)
        self.scale, self.stream_id, self.draws = 1.0, 0, []

    # -- the data-parallel surface DataParallelFQL uses
    def grad_tensor(self):
        return self._g

    def grad_buckets(self):
        return None

    def set_grad_scale(self, s):
        self.scale = s

    def set_rng_stream(self, sid):
        self.stream_id = sid

    def get_params(self):
        return self.ref.params

    def set_params(self, p):
        self.ref.params = O.tree_map(lambda a: np.asarray(a, dtype=np.float64), p)

    def get_opt_state(self):
        return {'count': self.ref.count, 'step': self.ref.step, 'mu': self.ref.mu, 'nu': self.ref.nu}

    def set_opt_state(self, s):
        self.ref.mu = O.tree_map(lambda a: np.asarray(a, dtype=np.float64), s['mu'])
        self.ref.nu = O.tree_map(lambda a: np.asarray(a, dtype=np.float64), s['nu'])
        self.ref.count, self.ref.step = s['count'], s['step']

    def upload_dataset(self, ds):
        self.ds, self.rows = ds, len(ds['observations'])

    def update_begin(self, idxs=None, shard=(0, 0), batch_size=None, noise=None, stream=None, batch=None):
        lo, hi = shard if shard != (0, 0) else (0, self.rows)
        rng = np.random.default_rng([self.config['seed'], self.stream_id, self.ref.count])   # engine: Philox(key ^ f(stream), step)
        idx = lo + rng.integers(0, hi - lo, size=batch_size) if idxs is None else np.asarray(idxs)
        nz = O.make_noise(batch_size, self.ad, int(rng.integers(1 << 30))) if noise is None else noise
        self.draws.append((idx, nz))
        _, self.info, g = self.ref.grads(O.sample_batch(self.ds, idx), nz)
        flat = dict(tree_flatten(g))
        self._g.copy_(torch.from_numpy(np.concatenate([flat[p].reshape(-1) for p, _, _ in self.layout])))

    def update_end(self, stream=None):
        g = (self._g * self.scale).numpy()
        out, o = [], 0
        for p, shp, n in self.layout:
            out.append((p, g[o:o + n].reshape(shp)))
            o += n
        for p, v in tree_flatten(self.ref.params):
            if 'target' in p:
                out.append((p, np.zeros_like(v)))
        self.ref.apply_gradients(tree_unflatten(out))


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from fql_amd.parallel import DataParallelFQL
    od, ad, B, N = 7, 3, 8, 64
    cfg = dict(O.get_config(), actor_hidden_dims=(16, 16), value_hidden_dims=(16, 16), alpha=10.0, seed=5)
    ds = O.make_synthetic_dataset(N, od, ad, seed=0)
    # different parameter seeds and optimizer histories per rank: the broadcast must erase the difference
    agent = _CpuAgent(100 + rank, od, ad, cfg, ds)
    agent.ref.count, agent.ref.step = 3 * rank, 1 + 3 * rank
    agent.ref.mu = O.tree_map(lambda a: a + 0.01 * (rank + 1), agent.ref.mu)
    try:
        DataParallelFQL(_CpuAgent(1, od, ad, dict(cfg, normalize_q_loss=True), ds))
        refused = False
    except ValueError as e:
        refused = 'normalize_q_loss' in str(e)
    dp = DataParallelFQL(agent)
    lo, hi = dp.upload_shard(ds)
    p0 = np.concatenate([v.reshape(-1) for _, v in tree_flatten(agent.ref.params)])
    m0 = np.concatenate([v.reshape(-1) for _, v in tree_flatten(agent.ref.mu)])
    state0 = (agent.ref.count, agent.ref.step, agent.stream_id, agent.scale, lo, hi, agent.rows)
    dp.update_from_dataset(batch_size=B)
    dp.update_from_dataset(batch_size=B)
    info = dp.reduce_info({'a/x': float(rank), 'a/q_max': float(rank), 'a/q_min': float(rank)})
    p2 = np.concatenate([v.reshape(-1) for _, v in tree_flatten(agent.ref.params)])
    q.put((rank, refused, p0, m0, state0, [(lo + 0 * i, nz['z']) for i, (idx, nz) in enumerate(agent.draws)], [d[0] for d in agent.draws], p2, info))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_wrapper_itself_under_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ref0, p0a, m0a, st_a, za, ia, p2a, info_a), (_, ref1, p0b, m0b, st_b, zb, ib, p2b, info_b) = res
    assert ref0 and ref1                                                    # normalize_q_loss is refused at W = 2
    np.testing.assert_array_equal(p0a, p0b)                                 # params broadcast from rank 0 ...
    np.testing.assert_array_equal(m0a, m0b)                                 # ... and the Adam moments
    assert st_a[:2] == st_b[:2] == (0, 1)                                   # ... and count / step (rank 0's)
    assert (st_a[2], st_b[2]) == (0, 1) and st_a[3] == st_b[3] == 0.5       # rank in the RNG stream; grad scale 1/W
    assert (st_a[4], st_a[5], st_b[4], st_b[5]) == (0, 32, 32, 64) and st_a[6] == st_b[6] == 32   # physical shards
    assert not np.array_equal(za[0][1], zb[0][1])                           # the two ranks draw different noise ...
    assert not np.array_equal(ia[0], ib[0])                                 # ... and different rows
    np.testing.assert_allclose(p2a, p2b, rtol=0, atol=1e-15)                # replicas stay identical after two steps
    assert np.abs(p2a - p0a).max() > 1e-5
    assert info_a == info_b == {'a/x': 0.5, 'a/q_max': 1.0, 'a/q_min': 0.0}


def test_two_rank_step_equals_one_rank_step_on_concatenated_batch_through_the_wrapper():
    """The same contract as above, but THROUGH DataParallelFQL: two _CpuAgent replicas (one process, world size 1 each would not
    reduce) are emulated by feeding one agent the concatenated draws of the 2-rank run and comparing post-step parameters."""
    od, ad, B, N = 7, 3, 8, 64
    cfg = dict(O.get_config(), actor_hidden_dims=(16, 16), value_hidden_dims=(16, 16), alpha=10.0, seed=5)
    ds = O.make_synthetic_dataset(N, od, ad, seed=0)
    ranks = [_CpuAgent(100, od, ad, cfg, ds) for _ in range(2)]
    idxs = [np.random.default_rng(r).integers(32 * r, 32 * (r + 1), size=B) for r in range(2)]
    noises = [O.make_noise(B, ad, 50 + r) for r in range(2)]
    for r in range(2):
        ranks[r].update_begin(idxs=idxs[r], batch_size=B, noise=noises[r])
    total = ranks[0].grad_tensor() + ranks[1].grad_tensor()                  # what the all-reduce leaves on every rank
    for r in range(2):
        ranks[r].grad_tensor().copy_(total)
        ranks[r].set_grad_scale(0.5)
        ranks[r].update_end()
    one = _CpuAgent(100, od, ad, cfg, ds)
    one.update_begin(idxs=np.concatenate(idxs), batch_size=2 * B, noise={k: np.concatenate([n[k] for n in noises]) for k in noises[0]})
    one.update_end()
    for (p, a), (_, b) in zip(tree_flatten(ranks[0].ref.params), tree_flatten(one.ref.params)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12, err_msg=p)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher (how the driver calls it): the parent, which never touches a GPU, starts N ranks with the
    torch.distributed environment, relays rank 0's single line and returns the worst exit code.  Here (no GPU) up to device selection:
    every rank stops at 'only 0 GPU(s) visible' and the parent reports failure; with the dry-run hook rank 0's environment comes back."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=dict(env, FQL_BENCH_SPAWN_DRYRUN='1'), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines           # ONE line on stdout: rank 0's
    got = json.loads(lines[0])
    assert got['RANK'] == '0' and got['WORLD_SIZE'] == '2' and got['MASTER_ADDR'] == '127.0.0.1' and got['gpus'] == 2
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0
        assert r.stderr.count('GPU(s) visible') == 2, r.stderr[-2000:]     # both ranks got as far as choosing their device
        assert r.stdout.strip() == ''
