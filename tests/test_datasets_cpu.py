"""CPU: host mirror of Dataset / ReplayBuffer (utils/datasets.py:36-100,435-495)."""
import numpy as np

from fql_amd.datasets import Dataset, ReplayBuffer
from oracle import fql_oracle as O


def test_dataset_sample_follows_numpy_legacy_stream_and_explicit_idxs():
    ds = Dataset.create(**O.make_synthetic_dataset(100, 5, 2, seed=0))
    assert ds.size == 100 and not ds['observations'].flags.writeable
    np.random.seed(7)
    want = np.random.randint(100, size=16)
    np.random.seed(7)
    b = ds.sample(16)
    np.testing.assert_array_equal(b['observations'], ds['observations'][want])    # utils/datasets.py:64-66,94-100
    idx = np.array([3, 3, 99, 0])
    b2 = ds.sample(4, idxs=idx)
    for k in ds:
        np.testing.assert_array_equal(b2[k], ds[k][idx])


def test_replay_buffer_ring_semantics_match_reference():
    init = O.make_synthetic_dataset(6, 5, 2, seed=1)
    rb = ReplayBuffer.create_from_initial_dataset(init, size=8)
    assert (rb.size, rb.pointer, rb.max_size) == (6, 6, 8)
    tr = {k: (np.ones(v.shape[1:], v.dtype) if v.ndim > 1 else v.dtype.type(1)) for k, v in init.items()}
    sizes = []
    for _ in range(4):
        rb.add_transition(tr)
        sizes.append((rb.size, rb.pointer))
    assert sizes == [(7, 7), (7, 0), (7, 1), (7, 2)]          # size = max(pointer, size): never reports 8 after a wrap
    assert rb['rewards'][6] == 1 and rb['rewards'][0] == 1 and rb['rewards'][2] != 1
    rb.clear()
    assert (rb.size, rb.pointer) == (0, 0)
    empty = ReplayBuffer.create({k: v[0] for k, v in init.items()}, size=4)
    assert empty.max_size == 4 and empty.size == 0 and empty['observations'].shape == (4, 5)
