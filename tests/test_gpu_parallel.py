"""GPU: the two halves of the data-parallel step (fql_update_begin / all-reduce / fql_update_end) on ONE GPU:
two engine replicas each take half of a batch, their gradient buffers are summed through zero-copy torch
views (what RCCL all_reduce does across ranks) and scaled by 1/2 in the optimizer kernel; the result must
equal one engine stepping on the concatenated batch (SURVEY.md 8e correctness test)."""
import numpy as np
import pytest
import torch

from oracle import fql_oracle as O
from tests.util import make_problem, randomize_params

pytestmark = pytest.mark.gpu


def test_two_replicas_equal_one_step_on_concatenated_batch():
    import fql_amd
    from fql_amd.parallel import _DevView
    od, ad, B = 29, 8, 32
    cfg, ds, batch, noise = make_problem(od, ad, 2 * B, (64, 64, 64, 64), seed=5)
    halves = [({k: v[i * B:(i + 1) * B] for k, v in batch.items()}, {k: v[i * B:(i + 1) * B] for k, v in noise.items()}) for i in range(2)]
    full = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    params = randomize_params(full.get_params(), seed=3)
    full.set_params(params)
    reps = []
    for i in range(2):
        c = dict(cfg); c['batch_size'] = B
        a = fql_amd.FQLAgent.create(i, batch['observations'][:1], batch['actions'][:1], c)
        a.set_params(params)
        a.set_grad_scale(0.5)
        reps.append(a)
    views = []
    for a in reps:
        ptr, n = a.grad_buffer()
        t = torch.as_tensor(_DevView(ptr, n), device='cuda')
        assert t.data_ptr() == ptr and t.numel() == n
        views.append(t)
    for a, (b, nz) in zip(reps, halves):
        a.update_begin(batch=b, noise=nz)
    torch.cuda.synchronize()
    total = views[0] + views[1]
    views[0].copy_(total); views[1].copy_(total)
    torch.cuda.synchronize()
    for a in reps:
        a.update_end()
    full.update(batch, noise=noise)
    want = dict(O.tree_leaves_with_path(full.get_params()))
    gfull = dict(O.tree_leaves_with_path(full.get_opt_state()['mu']))
    for a in reps:
        got = dict(O.tree_leaves_with_path(a.get_params()))
        gmu = dict(O.tree_leaves_with_path(a.get_opt_state()['mu']))
        for p in want:
            scale = np.abs(gfull[p]).max() + 1e-12
            np.testing.assert_allclose(gmu[p], gfull[p], rtol=0, atol=1e-5 * scale + 1e-10, err_msg=p)
            stable = np.abs(gfull[p]) > 1e-3 * scale
            assert np.abs(got[p] - want[p])[stable].max(initial=0) <= 2e-6, p
    # replicas stay bit-identical to each other
    a0, a1 = (dict(O.tree_leaves_with_path(r.get_params())) for r in reps)
    for p in a0:
        np.testing.assert_array_equal(a0[p], a1[p], err_msg=p)


def test_update_from_dataset_matches_update_with_same_rows():
    import fql_amd
    od, ad, B = 29, 8, 32
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=9)
    idx = np.random.default_rng(4).integers(0, len(ds['observations']), size=B)
    a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    b = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    b.set_params(a.get_params())
    a.upload_dataset(ds)
    assert a.dataset_size()[0] == len(ds['observations'])
    a.update_from_dataset(B, idxs=idx, noise=noise)
    b.update(O.sample_batch(ds, idx), noise=noise)
    ia, ib = a.read_info(), b.read_info()
    for k in ia:
        assert ia[k] == ib[k], k                     # same kernels, same inputs: bitwise equal
    # engine RNG path: runs, advances, stays finite; shard restricts the sampled rows
    for _ in range(3):
        a.update_from_dataset(B, shard=(0, 64))
    assert all(np.isfinite(v) for v in a.read_info().values())
    assert a.get_opt_state()['count'] == 4


def test_replay_ring_insert():
    import fql_amd
    od, ad = 5, 2
    cfg = fql_amd.get_config(); cfg.update(actor_hidden_dims=(32, 32), value_hidden_dims=(32, 32), batch_size=16)
    ds = O.make_synthetic_dataset(6, od, ad, seed=0)
    a = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
    a.upload_dataset(ds, capacity=8)
    assert a.dataset_size() == (6, 6)
    tr = dict(observations=np.ones(od, np.float32), actions=np.zeros(ad, np.float32), rewards=-1.0, masks=1.0,
              next_observations=np.ones(od, np.float32))
    for i in range(3):
        a.add_transition(tr)
    # utils/datasets.py:489-491 verbatim: pointer = (pointer + 1) % max_size; size = max(pointer, size).
    # 6 -> 7 -> wraps to pointer 0 (size stays 7, the reference never reports max_size after a wrap) -> 1.
    assert a.dataset_size() == (7, 1)


@pytest.mark.parametrize('precision,H', [('fp32', 64), ('bf16x3', 64), ('fp32', 256), ('bf16x3', 256)])
def test_split_lane_update_matches_plain_update(precision, H):
    """fql_update_begin_split (lane graphs on two streams, bucketed gradients) must give the same step as fql_update - in both precisions,
    and at a width where the Euler chain / tail dgrads run on the chain kernels (H = 256)."""
    import fql_amd
    od, ad, B = 29, 8, 64
    cfg, ds, batch, noise = make_problem(od, ad, B, (H, H, H, H), seed=23, precision=precision)
    a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    b = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    b.set_params(a.get_params())
    buckets = b.grad_buckets()
    assert buckets is not None and buckets[0][0] == 0 and buckets[0][1] == buckets[1][0]
    ptr, n = b.grad_buffer()
    assert buckets[1][0] + buckets[1][1] == n
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    # fp32: the two programs run the same fp32 fma chains in another tile shape / lane structure (LayerNorm partial sums fold in another order).
    # bf16x3: the two-lane program leaves some products on the 16-row fp32 kernel that the three-lane program runs on the split side tiles, so the
    # two agree to the split's own accuracy, not to rounding.
    gtol, itol = (2e-6, 1e-6) if precision == 'fp32' else (1e-4, 5e-5)
    for step in range(3):
        nz = O.make_noise(B, ad, 40 + step)
        a.update(batch, noise=nz)
        torch.cuda.synchronize()
        b.update_begin_split(s0.cuda_stream, s1.cuda_stream, batch=batch, noise=nz)
        s0.wait_stream(s1)
        b.update_end(stream=s0.cuda_stream)
        torch.cuda.synchronize()
        ia, ib = a.read_info(), b.read_info()
        for k in ia:
            assert abs(ia[k] - ib[k]) <= itol * max(1.0, abs(ia[k])), (step, k, ia[k], ib[k])
        if step == 0:
            # the GRADIENTS of the two programs (Adam's first moment after one step from zero moments = 0.1 g), leaf by leaf: a missing
            # cross-stream edge in the split program would show here as a wrong gradient, whereas post-Adam parameters only show +-lr
            ma, mb = (dict(O.tree_leaves_with_path(x.get_opt_state()['mu'])) for x in (a, b))
            for p in ma:
                scale = np.abs(ma[p]).max()
                np.testing.assert_allclose(mb[p], ma[p], rtol=0, atol=gtol * scale + 1e-12, err_msg=f'gradient {p}')
    pa, pb = (dict(O.tree_leaves_with_path(x.get_params())) for x in (a, b))
    for p in pa:   # three sign-like Adam steps: an element whose gradient rounds differently near 0 moves by at most 2 lr per step
        assert np.abs(pb[p] - pa[p]).max() <= 3 * 2 * cfg['lr'] + 1e-6, p


@pytest.mark.parametrize('program', ['begin_end', 'begin_split_end_split'])
def test_data_parallel_programs_match_oracle_at_benchmark_size(program):
    """The two programs the N > 1 bench runs - fql_update_begin + fql_update_end and fql_update_begin_split + fql_update_end_split - against the
    fp64 oracle at BASELINE configs[1]'s size (B 256, hidden 512 x 4): infos, every leaf's gradient, nu, post-step parameters."""
    import fql_amd
    from tests.util import assert_step_matches
    od, ad, B = 29, 8, 256
    cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=43)
    a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    params = randomize_params(a.get_params(), seed=9, scale=0.05)
    a.set_params(params)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def step(agent, bt, nz):
        if program == 'begin_end':
            agent.update_begin(batch=bt, noise=nz, stream=s0.cuda_stream)
            agent.update_end(stream=s0.cuda_stream)
        else:
            agent.update_begin_split(s0.cuda_stream, s1.cuda_stream, batch=bt, noise=nz)
            agent.update_end_split(s0.cuda_stream, s1.cuda_stream)
        torch.cuda.synchronize()
        return None

    worst = assert_step_matches(a, ref, cfg, batch, noise, step=step)
    assert worst[0] <= 2e-5, worst


def test_dataset_mirror_attached_to_engine():
    """fql_amd.datasets.ReplayBuffer.attach: host ring and device ring stay in step; sampling by explicit idxs on the
    device equals update(ds.sample(idxs)) on the host path."""
    import fql_amd
    from fql_amd.datasets import ReplayBuffer
    od, ad, B = 5, 2, 16
    cfg = fql_amd.get_config(); cfg.update(actor_hidden_dims=(32, 32), value_hidden_dims=(32, 32), batch_size=B, alpha=10.0)
    init = O.make_synthetic_dataset(40, od, ad, seed=3)
    a = fql_amd.FQLAgent.create(0, init['observations'][:1], init['actions'][:1], cfg)
    b = fql_amd.FQLAgent.create(0, init['observations'][:1], init['actions'][:1], cfg)
    b.set_params(a.get_params())
    rb = ReplayBuffer.create_from_initial_dataset(init, size=48).attach(a)
    extra = O.make_synthetic_dataset(12, od, ad, seed=4)
    for i in range(12):                                    # wraps the 48-row ring
        rb.add_transition({k: v[i] for k, v in extra.items()})
    assert a.dataset_size() == (rb.size, rb.pointer)
    idx = np.random.default_rng(0).integers(0, rb.size, size=B)
    nz = O.make_noise(B, ad, 5)
    a.update_from_dataset(B, idxs=idx, noise=nz)
    b.update(rb.sample(B, idxs=idx), noise=nz)
    assert a.read_info() == b.read_info()


@pytest.mark.parametrize('overlap', [True, False])
def test_data_parallel_wrapper_with_rccl_at_world_size_one(overlap):
    """DataParallelFQL exactly as bench.py drives it (torch's DEFAULT stream current, collectives issued through torch.distributed /
    RCCL even at world size 1): same step as the plain engine call, on the overlapped (bucketed) and the single-all-reduce path."""
    import os
    import torch.distributed as dist
    import fql_amd
    from fql_amd.parallel import DataParallelFQL
    od, ad, B = 29, 8, 64
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=27)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        created = True
    try:
        a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
        b = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
        a.upload_dataset(ds); b.upload_dataset(ds)
        dp = DataParallelFQL(b, overlap=overlap)
        dp.always_reduce = True
        assert (dp.buckets is not None) == overlap
        n = len(ds['observations'])
        for step in range(3):
            idxs = np.random.default_rng(step).integers(0, n, size=B)
            nz = O.make_noise(B, ad, 100 + step)
            a.update_from_dataset(B, idxs=idxs, noise=nz)
            dp.update_from_dataset(n, batch_size=B, idxs=idxs, noise=nz)
        torch.cuda.synchronize()
        ia, ib = a.read_info(), b.read_info()
        for k in ia:
            assert abs(ia[k] - ib[k]) <= 1e-6 * max(1.0, abs(ia[k])), (k, ia[k], ib[k])
        # second moments after three steps = 0.001 sum of 0.999^k g_k^2: a gradient-level comparison of the wrapper's step with the plain engine call
        # (the fused three-lane program uses 32 x 64 tiles throughout, the begin / end programs pick per launch: LayerNorm partial sums fold in
        # another order, so gradients agree to rounding, not bitwise; post-Adam parameters only show the sign-like +-lr of |g| ~ 0 elements)
        na, nb = (dict(O.tree_leaves_with_path(x.get_opt_state()['nu'])) for x in (a, b))
        for p in na:
            np.testing.assert_allclose(nb[p], na[p], rtol=0, atol=1e-5 * np.abs(na[p]).max() + 1e-15, err_msg=f'nu {p}')
        pa, pb = (dict(O.tree_leaves_with_path(x.get_params())) for x in (a, b))
        for p in pa:
            assert np.abs(pb[p] - pa[p]).max() <= 3 * 2 * cfg['lr'] + 1e-6, p
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('overlap,B,H', [(True, 32, 64), (False, 32, 64), (True, 256, 512)], ids=['overlapped', 'plain', 'overlapped-B256-H512'])
def test_two_processes_on_one_gpu_equal_one_step_on_the_concatenated_batch(tmp_path, overlap, B, H):
    """DataParallelFQL end to end with TWO real ranks (two processes, gloo, both on cuda:0): state broadcast from rank 0 (parameters,
    Adam moments, counters), physical shards, per-rank indices inside the shard, bucketed (overlapped) or single all-reduce,
    per-bucket Adam - against ONE engine stepping on the concatenated 2B batches from the same start."""
    import os
    import subprocess
    import sys
    import fql_amd
    od, ad = 29, 8
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(29600 + (os.getpid() % 200) + (50 if overlap else 0) + (25 if B > 32 else 0))
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'dp_worker.py'), str(r), port, str(tmp_path), str(int(overlap)), str(B), str(H)],
                              cwd=root) for r in range(2)]
    rcs = [p.wait(timeout=240) for p in procs]
    assert rcs == [0, 0], rcs
    # the reference: one engine, same start, batch = [rank 0's rows ; rank 1's rows] of the global dataset
    cfg, ds, _, _ = make_problem(od, ad, B, (H, H, H, H), seed=71)
    cfg2 = dict(cfg); cfg2['batch_size'] = 2 * B
    one = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
    one.set_params(randomize_params(one.get_params(), seed=72))
    one.update(O.sample_batch(ds, np.arange(B)), noise=O.make_noise(B, ad, 73))
    n = len(ds['observations'])
    for step in range(3):
        idx = np.random.default_rng(100 + step).integers(0, n // 2, size=2 * B)
        idx_global = np.concatenate([idx[:B], n // 2 + idx[B:]])          # rank 1's shard starts at n / 2
        one.update(O.sample_batch(ds, idx_global), noise=O.make_noise(2 * B, ad, 200 + step))
    want = dict(O.tree_leaves_with_path(one.get_params()))
    want_nu = dict(O.tree_leaves_with_path(one.get_opt_state()['nu']))
    got = [np.load(os.path.join(tmp_path, f'rank{r}.npz')) for r in range(2)]
    for p, w in want.items():
        k = p.replace('/', '|')
        np.testing.assert_array_equal(got[0][k], got[1][k], err_msg=p)     # replicas stay bit-identical
        # gradient level: Adam's second moments after the four steps (0.001 sum 0.999^k g_k^2) - the mean of two rank means against one mean over 2 B
        np.testing.assert_allclose(got[0]['nu|' + k], want_nu[p], rtol=0, atol=2e-5 * np.abs(want_nu[p]).max() + 1e-15, err_msg=f'nu {p}')
        d = np.abs(got[0][k] - w)
        if B <= 32:
            assert d.max() <= 2e-6, (p, d.max())
        else:   # 4.9 M parameters: the odd element with |g| ~ 0 takes a sign-like Adam step the other way (<= 2 lr per step)
            assert d.max() <= 4 * 2 * cfg['lr'] + 1e-6 and (d > 2e-6).sum() <= max(3, 1e-3 * d.size), (p, d.max(), int((d > 2e-6).sum()))
    x0, x1 = (np.load(os.path.join(tmp_path, f'xbc{r}.npy')) for r in range(2))
    assert not np.array_equal(x0, x1)                                      # different rows / noise per rank from the same seed
