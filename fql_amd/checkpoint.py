"""Checkpoint save / restore in the reference's layout (utils/flax_utils.py:162-202).

The reference pickles ``{'agent': flax.serialization.to_state_dict(agent)}`` to ``params_{epoch}.pkl``.  For
FQLAgent (agents/fql.py:18-20, utils/flax_utils.py:53-88) that state dict is

    {'rng': uint32[2],
     'network': {'step': int,
                 'params': {'modules_critic': {...}, 'modules_target_critic': {...},
                            'modules_actor_bc_flow': {...}, 'modules_actor_onestep_flow': {...}},
                 'opt_state': {'0': {'count': int32, 'mu': <params tree>, 'nu': <params tree>}, '1': {}}}}

(optax.adam = chain(scale_by_adam, scale): a 2-tuple of states, tuples serialise with string indices).
This module writes and reads exactly that nesting with numpy leaves.  SURVEY.md 8f N2.

**Parity unpinned**: no checkpoint written by the JAX reference exists in this environment (jax/flax are absent
and the reference ships none), so the layout above is restated from the reference's source, not verified against a
real file.  Two things are therefore handled tolerantly and REPORTED rather than assumed:

* visual agents register the BC-flow encoder twice (agents/fql.py:230-232: inside ``actor_bc_flow`` and as the
  separately callable ``actor_bc_flow_encoder``).  Whether flax stores it nested
  (``modules_actor_bc_flow/encoder``), top-level (``modules_actor_bc_flow_encoder``) or both depends on flax's
  module-sharing rules; ``from_state_dict`` accepts all three and returns which one it saw (``report['visual_layout']``).
  The engine keeps ONE BC-flow encoder (author intent: shared); when a file holds two different copies the nested
  one (the one the BC loss trains, agents/fql.py:58) is loaded and the other is listed under ``report['dropped']``.
* a file written by the reference holds ``jax.Array`` leaves, whose pickle stream calls
  ``jax._src.array._reconstruct_array(numpy_reconstruct, args, array_state, aval_state)``.  The loader maps that
  to the numpy reconstruction alone, so such a file loads WITHOUT jax installed.

Loading never executes code from the file: ``safe_load`` is a restricted unpickler that resolves only numpy's
array/dtype/scalar reconstructors, the jax array shim above, and builtin containers; anything else raises.
"""
from __future__ import annotations

import glob
import io
import os
import pickle
from typing import Any, Dict, List, Tuple

import numpy as np

BC_ENC_TOP = 'modules_actor_bc_flow_encoder'
BC_MOD = 'modules_actor_bc_flow'


# ---------------------------------------------------------------------------------------------------------
# restricted unpickler
# ---------------------------------------------------------------------------------------------------------
def _jax_reconstruct_array(fun, args, arr_state, aval_state=None):
    """jax._src.array._reconstruct_array without jax: the numpy half of it (the device_put / aval half is dropped)."""
    a = fun(*args)
    a.__setstate__(arr_state)
    return a


def _allowed_globals() -> Dict[Tuple[str, str], Any]:
    import collections
    table: Dict[Tuple[str, str], Any] = {
        ('builtins', 'dict'): dict, ('builtins', 'list'): list, ('builtins', 'tuple'): tuple, ('builtins', 'set'): set,
        ('builtins', 'frozenset'): frozenset, ('builtins', 'int'): int, ('builtins', 'float'): float,
        ('builtins', 'bool'): bool, ('builtins', 'str'): str, ('builtins', 'bytes'): bytes, ('builtins', 'complex'): complex,
        ('builtins', 'slice'): slice, ('builtins', 'bytearray'): bytearray,
        ('collections', 'OrderedDict'): collections.OrderedDict,
        ('numpy', 'ndarray'): np.ndarray, ('numpy', 'dtype'): np.dtype,
        ('jax._src.array', '_reconstruct_array'): _jax_reconstruct_array,
        ('jax.interpreters.xla', '_reconstruct_array'): _jax_reconstruct_array,   # older jax
    }
    # numpy moved its C helpers from numpy.core to numpy._core (2.x); pickles name either
    import numpy.core.multiarray as _ma   # noqa: WPS433 (alias module exists on 1.x and 2.x)
    for mod in ('numpy.core.multiarray', 'numpy._core.multiarray'):
        table[(mod, '_reconstruct')] = _ma._reconstruct
        table[(mod, 'scalar')] = _ma.scalar
    try:
        from numpy._core import numeric as _num
    except Exception:   # numpy 1.x
        from numpy.core import numeric as _num
    for mod in ('numpy.core.numeric', 'numpy._core.numeric'):
        table[(mod, '_frombuffer')] = _num._frombuffer
    for name in ('bool_', 'int8', 'int16', 'int32', 'int64', 'uint8', 'uint16', 'uint32', 'uint64', 'float16', 'float32',
                 'float64'):
        table[('numpy', name)] = getattr(np, name)
    return table


class _SafeUnpickler(pickle.Unpickler):
    _table = None

    def find_class(self, module, name):
        if _SafeUnpickler._table is None:
            _SafeUnpickler._table = _allowed_globals()
        try:
            return _SafeUnpickler._table[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(
                f'checkpoint refers to {module}.{name}, which the restricted loader does not resolve '
                '(only numpy arrays / scalars, jax arrays as numpy, and builtin containers are loaded)') from None


def safe_load(f) -> Any:
    """pickle.load restricted to numpy arrays/scalars, jax arrays (read as numpy) and builtin containers."""
    if isinstance(f, (bytes, bytearray)):
        f = io.BytesIO(f)
    return _SafeUnpickler(f).load()


# ---------------------------------------------------------------------------------------------------------
# state dict <-> engine
# ---------------------------------------------------------------------------------------------------------
def to_state_dict(agent, visual_layout: str = 'nested') -> dict:
    """flax.serialization.to_state_dict(agent) for the engine-backed FQLAgent.

    visual_layout (visual agents only): 'nested' = modules_actor_bc_flow/encoder (default), 'toplevel' = also a
    top-level modules_actor_bc_flow_encoder holding the same (shared) arrays, 'toplevel-only' = the encoder ONLY under
    modules_actor_bc_flow_encoder (no nested copy) - the layout flax is likely to produce, since ModuleDict adopts the shared encoder
    instance first (agents/fql.py:230-232); use it to export to the JAX reference (unverified: no reference checkpoint exists here)."""
    opt = agent.get_opt_state()
    rng = np.asarray(getattr(agent, 'rng', [(agent._seed >> 32) & 0xFFFFFFFF, agent._seed & 0xFFFFFFFF]), dtype=np.uint32)
    trees = [agent.get_params(), opt['mu'], opt['nu']]
    if visual_layout in ('toplevel', 'toplevel-only'):
        for t in trees:
            if isinstance(t.get(BC_MOD), dict) and 'encoder' in t[BC_MOD]:
                t[BC_ENC_TOP] = t[BC_MOD]['encoder']
                if visual_layout == 'toplevel-only':
                    t[BC_MOD] = {k: v for k, v in t[BC_MOD].items() if k != 'encoder'}
    elif visual_layout != 'nested':
        raise ValueError("visual_layout must be 'nested', 'toplevel' or 'toplevel-only'")
    return {
        'rng': rng,
        'network': {
            'step': np.int64(opt['step']),
            'params': trees[0],
            'opt_state': {'0': {'count': np.int32(opt['count']), 'mu': trees[1], 'nu': trees[2]}, '1': {}},
        },
    }


def _flatten(tree, prefix=''):
    out = []
    for k in sorted(tree):
        v = tree[k]
        p = f'{prefix}/{k}' if prefix else str(k)
        out.extend(_flatten(v, p) if isinstance(v, dict) else [(p, v)])
    return out


def _resolve_visual(tree: dict, report: dict, what: str) -> dict:
    """Map whichever BC-flow-encoder layout the file uses onto the engine's modules_actor_bc_flow/encoder."""
    tree = dict(tree)
    top = tree.pop(BC_ENC_TOP, None)
    bc = tree.get(BC_MOD)
    nested = bc.get('encoder') if isinstance(bc, dict) else None
    if top is None:
        layout = 'nested' if nested is not None else 'state'
    elif nested is None:
        layout = 'toplevel'
        tree[BC_MOD] = dict(bc or {}, encoder=top)
    else:
        same = all(np.array_equal(np.asarray(a), np.asarray(b)) for (_, a), (_, b) in zip(_flatten(top), _flatten(nested))) \
            and [p for p, _ in _flatten(top)] == [p for p, _ in _flatten(nested)]
        layout = 'both-shared' if same else 'both-distinct'
        if not same:
            report.setdefault('dropped', []).append(f'{what}:{BC_ENC_TOP} (distinct from {BC_MOD}/encoder, which is loaded)')
    prev = report.get('visual_layout')
    if prev is None or what == 'params':
        report['visual_layout'] = layout
    return tree


def from_state_dict(agent, state: dict, strict: bool = True) -> dict:
    """flax.serialization.from_state_dict(agent, state): loads params, Adam moments, count and step.

    Returns a report: {'visual_layout': 'state'|'nested'|'toplevel'|'both-shared'|'both-distinct', 'dropped': [...],
    'missing': [...], 'unexpected': [...]}.  With strict=True missing / unexpected leaves raise KeyError naming them."""
    report: Dict[str, Any] = {'dropped': [], 'missing': [], 'unexpected': []}
    net = state['network']
    want = {p for p, _ in agent.leaves()}

    def prepare(tree, what):
        tree = _resolve_visual(_np_tree(tree), report, what)
        have = {p for p, _ in _flatten(tree)}
        miss, extra = sorted(want - have), sorted(have - want)
        if what == 'params':
            report['missing'] += miss
            report['unexpected'] += extra
        if strict and (miss or extra):
            raise KeyError(f'checkpoint {what}: missing leaves {miss[:8]}{"..." if len(miss) > 8 else ""}, '
                           f'unexpected leaves {extra[:8]}{"..." if len(extra) > 8 else ""}')
        if extra:   # non-strict: ignore what the engine has no slot for
            keep = [(p, v) for p, v in _flatten(tree) if p in want]
            tree = {}
            for p, v in keep:
                node = tree
                ks = p.split('/')
                for k in ks[:-1]:
                    node = node.setdefault(k, {})
                node[ks[-1]] = v
        return tree

    agent.set_params(prepare(net['params'], 'params'))
    adam = net['opt_state']['0']
    agent.set_opt_state({'mu': prepare(adam['mu'], 'mu'), 'nu': prepare(adam['nu'], 'nu'), 'count': int(np.asarray(adam['count'])),
                         'step': int(np.asarray(net['step']))})
    if 'rng' in state and state['rng'] is not None:
        r = np.asarray(state['rng']).astype(np.uint64).reshape(-1)
        if r.size >= 2:
            agent.rng = r[:2].astype(np.uint32)
            agent._seed = int((int(r[0]) << 32) | int(r[1]))
    return report


def _np_tree(t):
    if isinstance(t, dict):
        return {str(k): _np_tree(v) for k, v in t.items()}
    return np.asarray(t, dtype=np.float32)


def save_agent(agent, save_dir, epoch, visual_layout: str = 'nested'):
    """utils/flax_utils.py:162-178."""
    save_dict = dict(agent=to_state_dict(agent, visual_layout))
    save_path = os.path.join(save_dir, f'params_{epoch}.pkl')
    with open(save_path, 'wb') as f:
        pickle.dump(save_dict, f)
    print(f'Saved to {save_path}')
    return save_path


def restore_agent(agent, restore_path, restore_epoch, strict: bool = True):
    """utils/flax_utils.py:181-202 (the glob must match exactly one directory).  The file is read with the restricted
    loader (`safe_load`); the layout report of `from_state_dict` is kept on ``agent.restore_report``."""
    candidates = glob.glob(restore_path)
    assert len(candidates) == 1, f'Found {len(candidates)} candidates: {candidates}'
    restore_path = candidates[0] + f'/params_{restore_epoch}.pkl'
    with open(restore_path, 'rb') as f:
        load_dict = safe_load(f)
    agent.restore_report = from_state_dict(agent, load_dict['agent'], strict=strict)
    print(f'Restored from {restore_path} (BC-flow encoder layout: {agent.restore_report["visual_layout"]})')
    return agent
