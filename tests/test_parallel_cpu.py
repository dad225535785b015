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
