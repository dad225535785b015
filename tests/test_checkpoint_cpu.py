"""CPU: the checkpoint module writes the reference's state-dict nesting (utils/flax_utils.py:162-202), loads it with a
restricted unpickler, and accepts every layout the BC-flow encoder of a visual agent may be stored in."""
import copy
import io
import pickle

import numpy as np
import pytest

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

    def leaves(self):
        return [(p, a.shape) for p, a in O.tree_leaves_with_path(self.p)]

    def get_params(self):
        return copy.deepcopy(self.p)

    def get_opt_state(self):
        return {'count': self.count, 'step': self.step, 'mu': copy.deepcopy(self.mu), 'nu': copy.deepcopy(self.nu)}

    def set_params(self, p):
        self.p = p

    def set_opt_state(self, s):
        self.mu, self.nu, self.count, self.step = s['mu'], s['nu'], s['count'], s['step']


def _small_cfg(**kw):
    return dict(O.get_config(), actor_hidden_dims=(8, 8), value_hidden_dims=(8, 8), **kw)


def _assert_same_tree(a, b):
    la, lb = O.tree_leaves_with_path(a), O.tree_leaves_with_path(b)
    assert [p for p, _ in la] == [p for p, _ in lb]
    for (p, x), (_, y) in zip(la, lb):
        np.testing.assert_array_equal(x, y, err_msg=p)


def test_state_dict_layout_and_pickle_roundtrip(tmp_path):
    cfg = _small_cfg()
    a = FakeAgent(O.init_params(0, 5, 2, cfg))
    path = checkpoint.save_agent(a, str(tmp_path), 3)
    assert path.endswith('params_3.pkl')
    with open(path, 'rb') as f:            # a file this test just wrote
        d = checkpoint.safe_load(f)
    st = d['agent']
    assert list(st['rng']) == [7, 9]
    assert set(st['network']) == {'step', 'params', 'opt_state'}
    assert set(st['network']['params']) == set(O.MODULES)
    assert int(st['network']['opt_state']['0']['count']) == 5 and st['network']['opt_state']['1'] == {}
    b = FakeAgent(O.init_params(1, 5, 2, cfg))
    b.count = b.step = 0
    checkpoint.restore_agent(b, str(tmp_path), 3)
    _assert_same_tree(a.p, b.p)
    _assert_same_tree(a.mu, b.mu)
    assert (b.count, b.step, b._seed) == (5, 6, (7 << 32) | 9)
    assert b.restore_report['visual_layout'] == 'state' and not b.restore_report['missing']


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ('echo pwned > /dev/null',))


def test_restricted_loader_refuses_code_execution():
    blob = pickle.dumps({'agent': {'x': _Evil()}})
    with pytest.raises(pickle.UnpicklingError, match='restricted loader'):
        checkpoint.safe_load(io.BytesIO(blob))
    # numpy containers of every kind a state dict holds do load
    ok = {'a': np.arange(6, dtype=np.float32).reshape(2, 3), 'b': np.int32(3), 'c': np.float32(1.5), 'd': [1, (2, 3)],
          'e': np.array([1, 2], dtype=np.uint32)}
    back = checkpoint.safe_load(pickle.dumps(ok))
    np.testing.assert_array_equal(back['a'], ok['a'])
    assert back['b'] == 3 and back['c'] == 1.5 and back['d'] == [1, (2, 3)]


class _FakeJaxArray:
    """Pickles exactly like jax.Array.__reduce__ does: (jax._src.array._reconstruct_array, (fun, args, arr_state, aval_state))."""

    def __init__(self, a):
        self.a = a

    def __reduce__(self):
        fun, args, arr_state = self.a.__reduce__()
        return (_reconstruct_array, (fun, args, arr_state, {'weak_type': False}))


def _reconstruct_array(*a):   # stands in for jax's function in the pickle stream (renamed below)
    raise AssertionError('never called: the loader must substitute its own shim')


def test_jax_array_leaves_load_as_numpy_without_jax():
    arr = np.arange(12, dtype=np.float32).reshape(3, 4)
    blob = pickle.dumps({'w': _FakeJaxArray(arr)}, protocol=4)
    # the stream names this test module; rewrite it to the module path jax uses (same length-prefixed opcode layout: re-frame by
    # unpickling with a find_class hook instead of byte surgery)
    class Rename(pickle.Unpickler):
        def find_class(self, module, name):
            if name == '_reconstruct_array':
                return checkpoint._SafeUnpickler(io.BytesIO(b'')).find_class('jax._src.array', '_reconstruct_array')
            return checkpoint._SafeUnpickler(io.BytesIO(b'')).find_class(module, name)
    back = Rename(io.BytesIO(blob)).load()
    assert isinstance(back['w'], np.ndarray)
    np.testing.assert_array_equal(back['w'], arr)


@pytest.mark.parametrize('layout', ['nested', 'toplevel', 'both-shared', 'both-distinct'])
def test_visual_bc_encoder_layouts(layout):
    cfg = _small_cfg(encoder='impala_small')
    params = O.init_params(0, (32, 32, 3), 2, cfg)
    a = FakeAgent(params)
    st = checkpoint.to_state_dict(a)
    assert 'encoder' in st['network']['params'][checkpoint.BC_MOD]
    trees = [st['network']['params'], st['network']['opt_state']['0']['mu'], st['network']['opt_state']['0']['nu']]
    for t in trees:
        enc = t[checkpoint.BC_MOD]['encoder']
        if layout == 'toplevel':
            t[checkpoint.BC_ENC_TOP] = enc
            t[checkpoint.BC_MOD] = {k: v for k, v in t[checkpoint.BC_MOD].items() if k != 'encoder'}
        elif layout == 'both-shared':
            t[checkpoint.BC_ENC_TOP] = copy.deepcopy(enc)
        elif layout == 'both-distinct':
            t[checkpoint.BC_ENC_TOP] = O.tree_map(lambda x: x + 1.0, enc)
    b = FakeAgent(O.init_params(1, (32, 32, 3), 2, cfg))
    rep = checkpoint.from_state_dict(b, pickle.loads(pickle.dumps(st)))
    assert rep['visual_layout'] == layout
    assert bool(rep['dropped']) == (layout == 'both-distinct')
    _assert_same_tree(a.p, b.p)
    _assert_same_tree(a.nu, b.nu)


def test_missing_and_unexpected_leaves_are_named():
    cfg = _small_cfg()
    a = FakeAgent(O.init_params(0, 5, 2, cfg))
    st = checkpoint.to_state_dict(a)
    del st['network']['params']['modules_critic']['value_net']['Dense_0']['bias']
    st['network']['params']['modules_extra'] = {'w': np.zeros(3, np.float32)}
    b = FakeAgent(O.init_params(1, 5, 2, cfg))
    with pytest.raises(KeyError, match='modules_critic/value_net/Dense_0/bias'):
        checkpoint.from_state_dict(b, st)


def test_toplevel_export_layout():
    cfg = _small_cfg(encoder='impala_small')
    a = FakeAgent(O.init_params(0, (32, 32, 3), 2, cfg))
    st = checkpoint.to_state_dict(a, visual_layout='toplevel')
    assert checkpoint.BC_ENC_TOP in st['network']['params'] and checkpoint.BC_ENC_TOP in st['network']['opt_state']['0']['mu']


def test_toplevel_only_export_round_trips():
    """'toplevel-only': the shared encoder ONLY under modules_actor_bc_flow_encoder (the layout flax is likely to write: ModuleDict adopts the
    shared instance first, agents/fql.py:230-232).  It must restore into the engine's tree unchanged."""
    cfg = _small_cfg(encoder='impala_small')
    a = FakeAgent(O.init_params(0, (32, 32, 3), 2, cfg))
    st = checkpoint.to_state_dict(a, visual_layout='toplevel-only')
    for tree in (st['network']['params'], st['network']['opt_state']['0']['mu'], st['network']['opt_state']['0']['nu']):
        assert checkpoint.BC_ENC_TOP in tree and 'encoder' not in tree[checkpoint.BC_MOD]
    b = FakeAgent(O.init_params(1, (32, 32, 3), 2, cfg))
    rep = checkpoint.from_state_dict(b, st)
    assert rep['visual_layout'] == 'toplevel' and not rep['missing']
    for (p, x), (_, y) in zip(O.tree_leaves_with_path(a.get_params()), O.tree_leaves_with_path(b.get_params())):
        np.testing.assert_array_equal(x, y, err_msg=p)
