"""GPU: online fine-tuning hooks - the replay ring beside the training dataset, balanced half/half sampling
(main.py:255-259), ring capacity for an uploaded dataset (create_from_initial_dataset, utils/datasets.py:457-473) and the uint8 ring
insert (utils/datasets.py:483-491).  The expected batches come from the host mirror of Dataset / ReplayBuffer (fql_amd/datasets.py,
itself checked against the reference's semantics in tests/test_datasets_cpu.py) and the CPU oracle."""
import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import assert_info_close, leaf_dict, make_problem, randomize_params

pytestmark = pytest.mark.gpu

from tests.test_gpu_hardening import _workspace  # noqa: E402


def _transition(rng, od, ad):
    return dict(observations=rng.standard_normal(od).astype(np.float32), actions=rng.uniform(-1, 1, ad).astype(np.float32),
                rewards=np.float32(rng.standard_normal()), terminals=np.float32(0), masks=np.float32(rng.random() < 0.9),
                next_observations=rng.standard_normal(od).astype(np.float32))


def test_balanced_update_equals_update_on_the_concatenated_halves_and_the_oracle():
    import fql_amd
    from fql_amd.datasets import Dataset, ReplayBuffer
    od, ad, B = 13, 4, 32
    cfg, ds, batch, noise = make_problem(od, ad, B, (64, 64, 64, 64), seed=21)
    a = fql_amd.FQLAgent.create(1, batch['observations'][:1], batch['actions'][:1], cfg)
    b = fql_amd.FQLAgent.create(1, batch['observations'][:1], batch['actions'][:1], cfg)
    params = randomize_params(a.get_params(), seed=22)
    a.set_params(params); b.set_params(params)
    train = Dataset.create(**{k: v.copy() for k, v in ds.items()})
    train.attach(a)
    with pytest.raises(ValueError):                       # no replay ring yet
        a.update_balanced(B)
    rb = ReplayBuffer.create({k: v[0] for k, v in train.items()}, size=20).attach(a, replay=True)
    with pytest.raises(ValueError):                       # empty ring: the reference's randint(0) raises too
        a.update_balanced(B)
    rng = np.random.default_rng(23)
    for _ in range(27):                                   # wraps: pointer 7; size = max(pointer, size) never reaches 20 (reference quirk)
        rb.add_transition(_transition(rng, od, ad))
    assert a.replay_size() == (rb.size, rb.pointer) == (19, 7)
    ia = rng.integers(0, train.size, size=B // 2); ib = rng.integers(0, rb.size, size=B // 2)
    ib[:3] = [0, 6, 18]                                   # overwritten rows and the last sampled row of the ring
    d, r = train.sample(B // 2, idxs=ia), rb.sample(B // 2, idxs=ib)
    cat = {k: np.concatenate([d[k], r[k]], axis=0) for k in ('observations', 'actions', 'rewards', 'masks', 'next_observations')}
    _, info_a = a.update_balanced(B, idxs=(ia, ib), noise=noise, want_info=True)
    _, info_b = b.update(cat, noise=noise)
    assert info_a == info_b                               # the same kernels on the same rows: bitwise equal
    for p, v in leaf_dict(b.get_params()).items():
        np.testing.assert_array_equal(leaf_dict(a.get_params())[p], v, err_msg=p)
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    _, info_r = ref.update(cat, noise)
    assert_info_close(info_a, info_r, rtol=5e-5, atol=5e-6)
    # engine-RNG draw: first half from the dataset rows, second half from the ring's `size` rows
    _, info_c = a.update_balanced(B, want_info=True)
    assert all(np.isfinite(v) for v in info_c.values())


def test_balanced_engine_rng_indices_stay_inside_their_source():
    """Rows [0, B/2) must index the dataset, rows [B/2, B) the ring's first `size` rows: marker values in the rewards prove it."""
    import fql_amd
    od, ad, B = 5, 2, 64
    cfg, ds, batch, _ = make_problem(od, ad, B, (32, 32), seed=31)
    a = fql_amd.FQLAgent.create(2, batch['observations'][:1], batch['actions'][:1], cfg)
    n = len(ds['observations'])
    ds = {k: v.copy() for k, v in ds.items()}
    ds['rewards'][:] = 1.0                                # dataset rows carry reward 1 ...
    a.upload_dataset(ds)
    a.create_replay_buffer(50)
    rng = np.random.default_rng(32)
    for i in range(9):                                    # ... ring rows 0..8 reward 2, the 41 unused rows stay 0
        t = _transition(rng, od, ad); t['rewards'] = np.float32(2.0)
        a.add_transition(t, replay=True)
    for _ in range(5):
        a.update_balanced(B)
        rew = _workspace(a, 5).reshape(-1)               # w_rew [B] of the last update (fql_debug_workspace)
        np.testing.assert_array_equal(rew[:B // 2], 1.0)
        np.testing.assert_array_equal(rew[B // 2:], 2.0)
    assert n > 0


def test_frames_ring_insert_and_balanced_frames_batch():
    """uint8 ring: reserve_dataset + add_transition on a frames dataset, then a balanced frames batch; expected batches from the host
    mirror's sample() (frame stacking with the reference's once-computed initial_locs) + the crop restatement."""
    import fql_amd
    from fql_amd.datasets import Dataset, ReplayBuffer
    from oracle import encoder_oracle as E
    from tests.test_gpu_visual import make_visual
    B, hw, ad, fs = 32, 32, 4, 3
    cfg, _, _ = make_visual(B=B, hw=hw, c=3 * fs, ad=ad)
    rng = np.random.default_rng(41)
    n = 40
    term = np.zeros(n, np.float32); term[[11, 25, 39]] = 1
    fields = dict(observations=rng.integers(0, 256, size=(n, hw, hw, 3), dtype=np.uint8),
                  next_observations=rng.integers(0, 256, size=(n, hw, hw, 3), dtype=np.uint8), terminals=term, masks=1 - term,
                  actions=rng.uniform(-1, 1, size=(n, ad)).astype(np.float32), rewards=-np.ones(n, np.float32))
    ex = np.zeros((1, hw, hw, 3 * fs), np.uint8)
    a = fql_amd.FQLAgent.create(7, ex, fields['actions'][:1], cfg)
    b = fql_amd.FQLAgent.create(7, ex, fields['actions'][:1], cfg)

    def frame_transition():
        return dict(observations=rng.integers(0, 256, size=(hw, hw, 3), dtype=np.uint8),
                    next_observations=rng.integers(0, 256, size=(hw, hw, 3), dtype=np.uint8), terminals=np.float32(0),
                    masks=np.float32(1), actions=rng.uniform(-1, 1, size=ad).astype(np.float32), rewards=np.float32(rng.standard_normal()))

    # (1) the dataset itself as the replay buffer (main.py:111-115): ring of 48 rows, 12 inserts wrap over rows 0..3
    ring = ReplayBuffer.create_from_initial_dataset(fields, size=48)
    ring.frame_stack = fs
    ring.attach(a)
    for _ in range(12):
        ring.add_transition(frame_transition())
    assert a.dataset_size() == (ring.size, ring.pointer) == (47, 4)
    idxs = rng.integers(0, ring.size, size=B); idxs[:6] = [0, 3, 4, 40, 41, 46]
    crops = rng.integers(0, 7, size=(B, 2))
    nz = O.make_noise(B, ad, 42)
    _, ia = a.update_from_dataset(B, idxs=idxs, noise=nz, want_info=True, crop_froms=crops)
    hb = ring.sample(B, idxs=idxs)
    hb['observations'] = E.random_crop_batch(hb['observations'], crops); hb['next_observations'] = E.random_crop_batch(hb['next_observations'], crops)
    _, ib = b.update(hb, noise=nz)
    assert ia == ib
    # (2) balanced: static frames dataset + a separate, initially empty frames ring
    train = Dataset.create(**{k: v.copy() for k, v in fields.items()}); train.frame_stack = fs
    train.attach(a)
    rb = ReplayBuffer.create({k: v[0] for k, v in fields.items()}, size=16); rb.frame_stack = fs
    rb.attach(a, replay=True)
    for _ in range(21):
        rb.add_transition(frame_transition())
    assert a.replay_size() == (rb.size, rb.pointer) == (15, 5)
    i0 = rng.integers(0, n, size=B // 2); i1 = rng.integers(0, rb.size, size=B // 2); i1[:3] = [0, 1, 14]
    d, r = train.sample(B // 2, idxs=i0), rb.sample(B // 2, idxs=i1)
    cat = {k: np.concatenate([d[k], r[k]], axis=0) for k in ('observations', 'actions', 'rewards', 'masks', 'next_observations')}
    cat['observations'] = E.random_crop_batch(cat['observations'], crops); cat['next_observations'] = E.random_crop_batch(cat['next_observations'], crops)
    nz2 = O.make_noise(B, ad, 43)
    _, ja = a.update_balanced(B, idxs=(i0, i1), noise=nz2, want_info=True, crop_froms=crops)
    _, jb = b.update(cat, noise=nz2)
    assert ja == jb
    _, jc = a.update_balanced(B, want_info=True)          # engine RNG: indices, one coin per half, offsets
    assert all(np.isfinite(v) for v in jc.values())


def test_reserve_turns_an_uploaded_state_dataset_into_a_ring_and_error_paths():
    """fql_dataset_reserve on a state dataset (ReplayBuffer.create_from_initial_dataset, main.py:111-115) + the refusals of the N4 entry points."""
    import fql_amd
    from fql_amd.datasets import ReplayBuffer
    od, ad, B = 7, 3, 16
    cfg, ds, batch, noise = make_problem(od, ad, B, (32, 32), seed=51)
    a = fql_amd.FQLAgent.create(3, batch['observations'][:1], batch['actions'][:1], cfg)
    b = fql_amd.FQLAgent.create(3, batch['observations'][:1], batch['actions'][:1], cfg)
    n = 40
    init = {k: v[:n].copy() for k, v in ds.items()}
    a.upload_dataset(init)                                  # capacity = n
    with pytest.raises(ValueError):
        a.reserve_dataset(n - 1)                            # cannot shrink
    a.reserve_dataset(48)
    assert a.dataset_size() == (n, n)
    rb = ReplayBuffer.create_from_initial_dataset(init, size=48)      # host mirror, not attached: the expected contents
    rng = np.random.default_rng(52)
    for _ in range(13):                                     # 8 fill rows 40..47, then 5 overwrite rows 0..4
        t = _transition(rng, od, ad)
        a.add_transition(t); rb.add_transition(t)
    assert a.dataset_size() == (rb.size, rb.pointer) == (47, 5)
    idx = np.array([0, 4, 5, 39, 40, 46] + list(rng.integers(0, 47, size=B - 6)))
    _, ia = a.update_from_dataset(B, idxs=idx, noise=noise, want_info=True)
    _, ib = b.update(rb.sample(B, idxs=idx), noise=noise)
    assert ia == ib
    # refusals
    with pytest.raises(ValueError):
        a.update_balanced(B)                                # no replay ring
    a.create_replay_buffer(4)
    with pytest.raises(ValueError):
        a.update_balanced(B)                                # empty ring
    a.add_transition(_transition(rng, od, ad), replay=True)
    a.update_balanced(B)                                    # one row is enough (every replay index is 0)
    np.testing.assert_array_equal(_workspace(a, 5).reshape(-1)[B // 2:], _workspace(a, 5).reshape(-1)[B // 2])
    with pytest.raises(ValueError):
        a.update_balanced(B, idxs=(np.zeros(B // 2, np.int64), np.zeros(3, np.int64)))   # wrong index count
    with pytest.raises(ValueError):
        a.add_transition(dict(_transition(rng, od, ad), observations=np.zeros((4, 4, 3), np.uint8)), replay=True)   # a frame into a state ring
