"""CPU: the checkpoint module writes the reference's state-dict nesting (utils/flax_utils.py:162-202)."""
import pickle

import numpy as np

from fql_amd import checkpoint
from oracle import fql_oracle as O


class FakeAgent:
    """Duck-types the slice of FQLAgent the checkpoint module touches."""

    def __init__(self, params):
        self.p = params
        self.mu = O.tree_map(lambda a: a * 0.1, params)
        self.nu = O.tree_map(lambda a: a * a, params)
        self.count, self.step, self._seed = 5, 6, (7 << 32) | 9
        self.rng = np.array([7, 9], dtype=np.uint32)

    def get_params(self):
        return self.p

    def get_opt_state(self):
        return {'count': self.count, 'step': self.step, 'mu': self.mu, 'nu': self.nu}

    def set_params(self, p):
        self.p = p

    def set_opt_state(self, s):
        self.mu, self.nu, self.count, self.step = s['mu'], s['nu'], s['count'], s['step']


def test_state_dict_layout_and_pickle_roundtrip(tmp_path):
    cfg = dict(O.get_config(), actor_hidden_dims=(8, 8), value_hidden_dims=(8, 8))
    a = FakeAgent(O.init_params(0, 5, 2, cfg))
    path = checkpoint.save_agent(a, str(tmp_path), 3)
    assert path.endswith('params_3.pkl')
    with open(path, 'rb') as f:            # a file this test just wrote
        d = pickle.load(f)
    st = d['agent']
    assert list(st['rng']) == [7, 9]
    assert set(st['network']) == {'step', 'params', 'opt_state'}
    assert set(st['network']['params']) == set(O.MODULES)
    assert int(st['network']['opt_state']['0']['count']) == 5 and st['network']['opt_state']['1'] == {}
    b = FakeAgent(O.init_params(1, 5, 2, cfg))
    b.count = b.step = 0
    checkpoint.restore_agent(b, str(tmp_path), 3)
    for (p, x), (_, y) in zip(O.tree_leaves_with_path(a.p), O.tree_leaves_with_path(b.p)):
        np.testing.assert_array_equal(x, y, err_msg=p)
    assert (b.count, b.step, b._seed) == (5, 6, (7 << 32) | 9)
