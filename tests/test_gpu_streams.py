"""GPU: stream ordering at the boundary, the per-rank RNG stream, and refusal (not a crash) of lane placements the HIP runtime
cannot capture.

* torch CUDA tensors produced on torch's DEFAULT stream are consumed correctly: the Python mirror passes that stream as
  FQL_STREAM_LEGACY (its handle reads 0 = "no stream" at the C ABI) and the engine enqueues its graph on it, so the batch is
  ordered behind the kernels that produce it and results are ordered before whatever the caller enqueues next;
* fql_set_rng_stream: same seed + different stream ids = different draws, same ids = identical draws;
* FQL_LANE_* overrides that make two forked lanes wait on each other in both directions raise ValueError (FQL_E_INVALID): the
  runtime bundled with torch recurses forever in hipStreamEndCapture on such a capture.
"""
import os

import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import make_problem, randomize_params

pytestmark = pytest.mark.gpu


def test_cuda_tensor_batch_and_observations_on_the_default_stream_match_the_numpy_path():
    import torch
    import fql_amd
    od, ad, B = 29, 8, 256
    cfg, ds, batch, noise = make_problem(od, ad, B, (512, 512, 512, 512), seed=3)
    a_np = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    a_cu = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    params = randomize_params(a_np.get_params(), seed=5, scale=0.05)
    a_np.set_params(params); a_cu.set_params(params)
    assert torch.cuda.current_stream().cuda_stream == 0                    # the default stream: handle 0
    dev = torch.device('cuda')
    for it in range(3):
        # the batch is PRODUCED on the default stream right before the call (a long-running producer in front of it): without
        # stream ordering the engine would read it half-written
        junk = torch.randn(4096, 4096, device=dev)
        for _ in range(4):
            junk = junk @ junk * 1e-3
        scale = 1.0 + 0.25 * it
        tb = {k: (torch.from_numpy(v).to(dev) * (scale if k.endswith('observations') else 1.0) + junk[0, 0] * 0.0)
              for k, v in batch.items() if k != 'terminals'}
        tn = {k: torch.from_numpy(v).to(dev) for k, v in noise.items()}
        nb = dict(batch, observations=batch['observations'] * np.float32(scale), next_observations=batch['next_observations'] * np.float32(scale))
        _, i_cu = a_cu.update(tb, noise=tn)
        _, i_np = a_np.update(nb, noise=noise)
        for k in O.INFO_KEYS:
            assert i_cu[k] == i_np[k], (it, k, i_cu[k], i_np[k])         # same kernels, same inputs: bitwise
        # sample_actions with CUDA observations returns a CUDA tensor that is ordered on the default stream
        obs = tb['observations'][:64]
        z = tn['z'][:64]
        out = a_cu.sample_actions(obs, noises=z)
        assert out.is_cuda
        ref = a_np.sample_actions(nb['observations'][:64], noises=noise['z'][:64])
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        fl = a_cu.compute_flow_actions(obs, z)
        np.testing.assert_array_equal(fl.cpu().numpy(), a_np.compute_flow_actions(nb['observations'][:64], noise['z'][:64]))
    # a NON-default torch stream is passed through as is
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tb = {k: torch.from_numpy(v).to(dev, non_blocking=True) for k, v in batch.items() if k != 'terminals'}
        loss_cu, _ = a_cu.total_loss(tb, noise={k: torch.from_numpy(v).to(dev) for k, v in noise.items()})
    loss_np, _ = a_np.total_loss(batch, noise=noise)
    assert loss_cu == loss_np


def _first_rows(agent, B, od):
    from tests.test_gpu_hardening import _draws
    noise, rows, _ = _draws(agent, B, od, 8)
    return noise['z'].copy(), rows.astype(np.int64)


def test_rng_stream_separates_replicas_created_with_the_same_seed():
    import fql_amd
    od, ad, B, N = 5, 8, 256, 4096
    cfg = fql_amd.get_config()
    cfg.update(actor_hidden_dims=(32, 32, 32), value_hidden_dims=(32, 32), alpha=10.0, batch_size=B)
    ds = O.make_synthetic_dataset(N, od, ad, seed=0)
    ds['observations'][:, 0] = np.arange(N, dtype=np.float32)
    out = []
    for stream_id in (0, 1, 1, 7):
        a = fql_amd.FQLAgent.create(3, ds['observations'][:1], ds['actions'][:1], cfg)   # the SAME seed on every "rank"
        a.upload_dataset(ds)
        a.set_rng_stream(stream_id)
        a.update_from_dataset(B)
        out.append(_first_rows(a, B, od))
        a.close()
    (z0, i0), (z1, i1), (z1b, i1b), (z7, i7) = out
    np.testing.assert_array_equal(z1, z1b); np.testing.assert_array_equal(i1, i1b)
    for za, ia, zb, ib in ((z0, i0, z1, i1), (z0, i0, z7, i7), (z1, i1, z7, i7)):
        assert not np.array_equal(za, zb) and not np.array_equal(ia, ib)
        assert abs(np.corrcoef(za.ravel(), zb.ravel())[0, 1]) < 0.1


def test_uncapturable_lane_placement_is_refused_not_crashed():
    import fql_amd
    cfg, ds, batch, noise = make_problem(7, 3, 16, (32, 32, 32, 32), seed=1)
    # one-step actor on lane 2, its consumer (critic on the actor's actions) on lane 1, and lane 2's Adam waiting on lane 1:
    # lanes 1 and 2 then wait on each other in both directions
    os.environ['FQL_LANE_os'] = '2'
    try:
        with pytest.raises(ValueError, match='both directions'):
            fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    finally:
        del os.environ['FQL_LANE_os']
    agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)   # the default three-lane program
    agent.update(batch, noise=noise)
    assert agent.stats()['launches_per_update'] > 0


def test_threaded_eager_lane_executor_passes_the_parity_cases():
    """FQL_NO_GRAPH=3 (run_threaded: one host thread per extra lane, cross-lane order through per-launch sequence numbers + events, nothing
    captured): the same parity cases as the captured graph, in a process of its own (the mode is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FQL_NO_GRAPH='3')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_parity.py', '-q', '-m', 'gpu', '-x', '-k',
                        'test_total_loss_and_update_match_oracle or test_full_size_config_one_update'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout
