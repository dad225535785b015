"""LazyInfo (fql_amd/agent.py): the info dict `update` returns.  It must hold its 13 keys for C-level consumers (json, pickle, len) before
anybody has read it through Python, and its values must behave as floats that fetch on first use.  CPU test with a stand-in engine."""
import ctypes as C
import json
import pickle

import numpy as np

from fql_amd.agent import INFO_KEYS, LazyInfo, LazyScalar


class _FakeLib:
    def __init__(self):
        self.waits = 0

    def fql_info_wait(self, h, ticket, buf):
        self.waits += 1
        for i in range(len(INFO_KEYS)):
            buf[i] = float(ticket) + 0.5 * i
        return 0


class _FakeAgent:
    def __init__(self):
        self._lib, self._h = _FakeLib(), None

    def _check(self, rc):
        assert rc == 0


def test_lazy_info_holds_13_keys_before_it_is_read():
    ag = _FakeAgent()
    info = LazyInfo(ag, 7)
    assert ag._lib.waits == 0
    assert len(info) == 13 and list(info) == list(INFO_KEYS)          # no fetch needed to see the keys
    assert ag._lib.waits == 0
    assert isinstance(info['critic/critic_loss'], LazyScalar)
    s = json.dumps(info, default=float)                                  # the C encoder walks the dict itself: all 13 keys are there
    assert json.loads(s) == {k: 7.0 + 0.5 * i for i, k in enumerate(INFO_KEYS)}
    assert ag._lib.waits == 1                                            # one fetch for all 13 values
    assert json.loads(json.dumps(info.to_dict())) == json.loads(s)
    back = pickle.loads(pickle.dumps(info))
    assert type(back) is dict and back == info.to_dict() and all(type(v) is float for v in back.values())
    try:
        json.dumps(LazyInfo(_FakeAgent(), 1))                            # like a dict of JAX arrays: loud, never a silent '{}'
        raise AssertionError('expected TypeError')
    except TypeError:
        pass


def test_lazy_scalar_behaves_as_a_float():
    info = LazyInfo(_FakeAgent(), 2)
    v = info['critic/q_mean']                                            # index 1 -> 2.5
    assert float(v) == 2.5 and v == 2.5 and v < 3 and 2 < v and abs(-v) == 2.5
    assert v + 1 == 3.5 and 1 + v == 3.5 and v * 2 == 5.0 and 5 / v == 2.0 and round(v, 0) == 2.0
    assert f'{v:.2f}' == '2.50' and np.isfinite(v) and float(np.asarray(v)) == 2.5
    other = LazyInfo(_FakeAgent(), 2)
    assert info == other and info.to_dict() == other.to_dict()


def test_host_gather_indices_are_checked_against_the_rows_on_the_device():
    """update_from_dataset / update_balanced / update_begin hand indices to a device gather: a host index outside [0, rows) must be refused
    (fql_amd/agent.py _host_idx), not turned into a zero row or an out-of-bounds read."""
    import pytest
    from fql_amd.agent import FQLAgent
    keep, ptr = FQLAgent._host_idx(None, [0, 5, 9], 3, 10, 'idxs')
    assert keep.dtype == np.int64 and ptr == keep.ctypes.data
    for bad in ([0, 5, 10], [-1, 2, 3]):
        with pytest.raises(ValueError, match='out of range'):
            FQLAgent._host_idx(None, bad, 3, 10, 'idxs')
    with pytest.raises(ValueError, match='must have 4 entries'):
        FQLAgent._host_idx(None, [0, 1, 2], 4, 10, 'idxs')
    import torch
    keep, ptr = FQLAgent._host_idx(None, torch.tensor([1, 2, 3]), 3, 10, 'idxs')
    assert ptr == keep.data_ptr()
    with pytest.raises(ValueError, match='out of range'):
        FQLAgent._host_idx(None, torch.tensor([1, 2, 30]), 3, 10, 'idxs')
