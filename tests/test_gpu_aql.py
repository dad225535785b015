"""GPU: the single-GPU update on the engine's own AQL queues (fql_amd/csrc/fql_aql.h) against the same update as a captured graph on a
HIP stream.  Same kernels, same arguments, same program order: the results must be bitwise equal, whatever is interleaved."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import make_problem, randomize_params, leaf_dict

pytestmark = pytest.mark.gpu


def _pair(B=64, H=128, seed=11):
    import fql_amd
    cfg, ds, batch, noise = make_problem(29, 8, B, (H,) * 4, seed=seed)
    agents = []
    for _ in range(2):
        a = fql_amd.FQLAgent.create(3, batch['observations'][:1], batch['actions'][:1], cfg)
        a.set_params(randomize_params(a.get_params(), seed + 1))
        a.upload_dataset(ds)
        agents.append(a)
    return cfg, ds, batch, agents


def _assert_same_state(a, b):
    pa, pb = leaf_dict(a.get_params()), leaf_dict(b.get_params())
    for k in pa:
        np.testing.assert_array_equal(pa[k], pb[k], err_msg=k)
    oa, ob = a.get_opt_state(), b.get_opt_state()
    assert oa['count'] == ob['count'] and oa['step'] == ob['step']
    for part in ('mu', 'nu'):
        la, lb = leaf_dict(oa[part]), leaf_dict(ob[part])
        for k in la:
            np.testing.assert_array_equal(la[k], lb[k], err_msg=f'{part} {k}')


def test_aql_update_equals_graph_update_bitwise():
    import torch
    cfg, ds, batch, (a, b) = _pair()
    B = cfg['batch_size']
    st = torch.cuda.Stream()
    rng = np.random.default_rng(5)
    for step in range(3):
        idxs = rng.integers(0, len(ds['observations']), size=B)
        nz = O.make_noise(B, 8, 40 + step)
        a.update_from_dataset(B, idxs=idxs, noise=nz)                          # stream left to the engine -> its own queues
        b.update_from_dataset(B, idxs=idxs, noise=nz, stream=st.cuda_stream)    # a caller's stream -> the captured graph
        assert a.synchronize() == 'aql'
        assert b.synchronize() == 'graph'
        st.synchronize()
        ia, ib = a.read_info(), b.read_info()
        assert ia == ib
    _assert_same_state(a, b)


def test_aql_many_updates_in_flight_and_hip_calls_in_between():
    """60 stream-less updates back to back (more than the 6 the queues hold in flight, more than the 8 signal sets), with calls that
    use HIP on the engine's buffers in between: each of those must first wait for the queues, each later update for the stream."""
    import torch
    cfg, ds, batch, (a, b) = _pair()
    B = cfg['batch_size']
    st = torch.cuda.Stream()
    noise = O.make_noise(B, 8, 9)
    for rnd in range(3):
        for _ in range(20):
            a.update_from_dataset(B)                                    # engine RNG: rows and noise from the device-side counters
            b.update_from_dataset(B, stream=st.cuda_stream)
        la, _ = a.total_loss(batch, noise=noise)                        # captured graph on the engine's HIP stream: after the queues
        st.synchronize()
        lb, _ = b.total_loss(batch, noise=noise)
        assert la == lb
        act_a = a.sample_actions(batch['observations'][:5], noises=noise['eps2'][:5])
        act_b = b.sample_actions(batch['observations'][:5], noises=noise['eps2'][:5])
        np.testing.assert_array_equal(act_a, act_b)
        a.update(batch, noise=noise)                                    # host batch: staging copies on the HIP stream, then the queues
        b.update(batch, noise=noise)
        assert a.synchronize() == 'aql'
    st.synchronize()
    assert a.read_info() == b.read_info()
    _assert_same_state(a, b)
    assert a.get_opt_state()['count'] == 63


def test_aql_can_be_switched_off():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import fql_amd\n"
            "from tests.util import make_problem\n"
            "cfg, ds, batch, noise = make_problem(29, 8, 32, (64,) * 4, seed=1)\n"
            "a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)\n"
            "a.upload_dataset(ds); a.update_from_dataset(32); print('MODE', a.synchronize())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, FQL_AQL='0'), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'MODE graph' in out.stdout


def test_aql_survives_batch_size_changes_balanced_sampling_and_teardown_with_updates_in_flight():
    """The engine's queues across the calls that rebuild or retire its programs: a batch-size change (new workspace, new packets, same queues and signals),
    balanced sampling (its own source descriptor), and an agent dropped while updates are still on the queues."""
    import torch
    import fql_amd
    from fql_amd.datasets import Dataset, ReplayBuffer
    od, ad = 13, 4
    cfg, ds, batch, noise = make_problem(od, ad, 32, (64,) * 4, seed=31)
    st = torch.cuda.Stream()
    agents = []
    for _ in range(2):
        a = fql_amd.FQLAgent.create(2, batch['observations'][:1], batch['actions'][:1], cfg)
        a.set_params(randomize_params(a.get_params(), 32))
        train = Dataset.create(**{k: v.copy() for k, v in ds.items()})
        train.attach(a)
        rb = ReplayBuffer.create({k: v[0] for k, v in train.items()}, size=64).attach(a, replay=True)
        rng = np.random.default_rng(33)
        for _ in range(40):
            rb.add_transition({'observations': rng.normal(size=od).astype(np.float32), 'actions': rng.uniform(-1, 1, size=ad).astype(np.float32),
                               'rewards': np.float32(-1.0), 'masks': np.float32(1.0), 'terminals': np.float32(0.0),
                               'next_observations': rng.normal(size=od).astype(np.float32)})
        agents.append(a)
    a, b = agents
    kw = dict(stream=st.cuda_stream)
    for B in (32, 64, 32):
        for _ in range(7):
            a.update_from_dataset(B)
            b.update_from_dataset(B, **kw)
        for _ in range(3):
            a.update_balanced(B)
            b.update_balanced(B, **kw)
        assert a.synchronize() == 'aql' and b.synchronize() == 'graph'
        st.synchronize()
        assert a.read_info() == b.read_info()
    _assert_same_state(a, b)
    for _ in range(5):
        a.update_from_dataset(32)          # still on the queues when the agent goes away: fql_destroy waits for them
    del a, agents
    import gc
    gc.collect()
    assert all(np.isfinite(v) for v in b.read_info().values())
