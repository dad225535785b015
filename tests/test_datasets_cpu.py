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


def test_frame_stack_and_augment_follow_the_reference():
    """utils/datasets.py:73-112 on the host mirror, against the oracle's restatement (oracle/encoder_oracle.py)."""
    from oracle import encoder_oracle as E
    from fql_amd.datasets import Dataset
    rng = np.random.default_rng(3)
    n = 60
    frames = rng.integers(0, 256, size=(n, 8, 8, 3), dtype=np.uint8)
    nxt = rng.integers(0, 256, size=(n, 8, 8, 3), dtype=np.uint8)
    term = np.zeros(n, np.float32); term[[9, 29, 59]] = 1
    ds = Dataset.create(observations=frames, next_observations=nxt, terminals=term, masks=1 - term,
                        actions=rng.uniform(-1, 1, size=(n, 2)).astype(np.float32), rewards=np.zeros(n, np.float32))
    np.testing.assert_array_equal(ds.initial_locs, [0, 10, 30])            # the last terminal starts no episode
    ds.frame_stack = 3
    idxs = np.array([0, 1, 10, 11, 12, 30, 59])
    b = ds.sample(len(idxs), idxs=idxs)
    obs, nobs = E.stack_frames(frames, nxt, term, idxs, 3)
    np.testing.assert_array_equal(b['observations'], obs)
    np.testing.assert_array_equal(b['next_observations'], nobs)
    assert b['observations'].shape == (7, 8, 8, 9)
    np.testing.assert_array_equal(b['observations'][2, ..., 0:3], frames[10])   # clamped to the episode start
    np.testing.assert_array_equal(b['observations'][2, ..., 6:9], frames[10])
    crops = rng.integers(0, 7, size=(7, 2))
    want_o, want_n = E.random_crop_batch(obs, crops), E.random_crop_batch(nobs, crops)
    ds.augment(b, ['observations', 'next_observations'], crop_froms=crops)
    np.testing.assert_array_equal(b['observations'], want_o)
    np.testing.assert_array_equal(b['next_observations'], want_n)
    ds.p_aug = 1.0
    np.random.seed(0)
    b2 = ds.sample(5)
    assert b2['observations'].shape == (5, 8, 8, 9) and b2['observations'].dtype == np.uint8


class _FakeAgent:
    """Records what the host mirror asks of the engine (fql_amd/datasets.py attach / add_transition paths)."""

    def __init__(self):
        self.calls = []

    def upload_dataset(self, ds, **kw):
        self.calls.append(('upload', len(ds['observations']), kw))

    def reserve_dataset(self, cap):
        self.calls.append(('reserve', cap))

    def create_replay_buffer(self, size):
        self.calls.append(('replay_create', size))

    def add_transition(self, tr, replay=False):
        self.calls.append(('add', bool(replay), float(tr['rewards'])))


def test_replay_buffer_attach_modes_drive_the_engine_as_main_py_does():
    """main.py:111-115 (the dataset IS the ring) and :106-109 (a separate, initially empty ring for balanced sampling)."""
    init = O.make_synthetic_dataset(6, 5, 2, seed=1)
    tr = {k: (np.ones(v.shape[1:], v.dtype) if v.ndim > 1 else v.dtype.type(3)) for k, v in init.items()}
    ag = _FakeAgent()
    rb = ReplayBuffer.create_from_initial_dataset(init, size=8).attach(ag)
    rb.add_transition(tr)
    assert ag.calls == [('upload', 6, {'capacity': 8}), ('add', False, 3.0)]
    ag2 = _FakeAgent()
    rb2 = ReplayBuffer.create({k: v[0] for k, v in init.items()}, size=4)
    rb2.add_transition(tr)                                   # a row held before attaching is replayed into the device ring
    rb2.attach(ag2, replay=True)
    rb2.add_transition(tr)
    assert ag2.calls == [('replay_create', 4), ('add', True, 3.0), ('add', True, 3.0)]
    assert (rb2.size, rb2.pointer) == (2, 2)
    # frames: upload the rows held, then grow the device ring to max_size
    n, hw = 5, 8
    fr = dict(observations=np.zeros((n, hw, hw, 3), np.uint8), next_observations=np.zeros((n, hw, hw, 3), np.uint8),
              terminals=np.zeros(n, np.float32), masks=np.ones(n, np.float32), actions=np.zeros((n, 2), np.float32), rewards=np.zeros(n, np.float32))
    ag3 = _FakeAgent()
    r3 = ReplayBuffer.create_from_initial_dataset(fr, size=9); r3.frame_stack = 3; r3.p_aug = 0.5
    r3.attach(ag3)
    assert ag3.calls == [('upload', 5, {'frame_stack': 3, 'p_aug': 0.5}), ('reserve', 9)]
