"""Checkpoint save / restore in the reference's layout (utils/flax_utils.py:162-202).

The reference pickles ``{'agent': flax.serialization.to_state_dict(agent)}`` to ``params_{epoch}.pkl``.  For
FQLAgent (agents/fql.py:18-20, utils/flax_utils.py:53-88) that state dict is

    {'rng': uint32[2],
     'network': {'step': int,
                 'params': {'modules_critic': {...}, 'modules_target_critic': {...},
                            'modules_actor_bc_flow': {...}, 'modules_actor_onestep_flow': {...}},
                 'opt_state': {'0': {'count': int32, 'mu': <params tree>, 'nu': <params tree>}, '1': {}}}}

(optax.adam = chain(scale_by_adam, scale): a 2-tuple of states, tuples serialise with string indices).
This module writes and reads exactly that nesting with numpy leaves, so a checkpoint written by the JAX
reference can seed this engine and vice versa.  SURVEY.md 8f N2.  The pickle is only ever *loaded* from paths
the caller names (``restore_agent``); nothing here touches files shipped with the reference.
"""
from __future__ import annotations

import glob
import os
import pickle

import numpy as np


def to_state_dict(agent) -> dict:
    """flax.serialization.to_state_dict(agent) for the engine-backed FQLAgent."""
    opt = agent.get_opt_state()
    rng = np.asarray(getattr(agent, 'rng', [(agent._seed >> 32) & 0xFFFFFFFF, agent._seed & 0xFFFFFFFF]), dtype=np.uint32)
    return {
        'rng': rng,
        'network': {
            'step': np.int64(opt['step']),
            'params': agent.get_params(),
            'opt_state': {'0': {'count': np.int32(opt['count']), 'mu': opt['mu'], 'nu': opt['nu']}, '1': {}},
        },
    }


def from_state_dict(agent, state: dict):
    """flax.serialization.from_state_dict(agent, state): loads params, Adam moments, count and step."""
    net = state['network']
    agent.set_params(_np_tree(net['params']))
    adam = net['opt_state']['0']
    agent.set_opt_state({'mu': _np_tree(adam['mu']), 'nu': _np_tree(adam['nu']), 'count': int(adam['count']),
                         'step': int(net['step'])})
    if 'rng' in state and state['rng'] is not None:
        r = np.asarray(state['rng']).astype(np.uint64).reshape(-1)
        if r.size >= 2:
            agent.rng = r[:2].astype(np.uint32)
            agent._seed = int((int(r[0]) << 32) | int(r[1]))
    return agent


def _np_tree(t):
    if isinstance(t, dict):
        return {k: _np_tree(v) for k, v in t.items()}
    return np.asarray(t, dtype=np.float32)


def save_agent(agent, save_dir, epoch):
    """utils/flax_utils.py:162-178."""
    save_dict = dict(agent=to_state_dict(agent))
    save_path = os.path.join(save_dir, f'params_{epoch}.pkl')
    with open(save_path, 'wb') as f:
        pickle.dump(save_dict, f)
    print(f'Saved to {save_path}')
    return save_path


def restore_agent(agent, restore_path, restore_epoch):
    """utils/flax_utils.py:181-202 (the glob must match exactly one directory)."""
    candidates = glob.glob(restore_path)
    assert len(candidates) == 1, f'Found {len(candidates)} candidates: {candidates}'
    restore_path = candidates[0] + f'/params_{restore_epoch}.pkl'
    with open(restore_path, 'rb') as f:
        load_dict = pickle.load(f)
    agent = from_state_dict(agent, load_dict['agent'])
    print(f'Restored from {restore_path}')
    return agent
