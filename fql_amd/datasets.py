"""Host mirror of the reference's batch source for state-based FQL (utils/datasets.py:36-112,435-495).

``Dataset`` / ``ReplayBuffer`` keep the reference's API (create, sample, get_random_idxs, get_subset,
create_from_initial_dataset, add_transition, clear, .size/.pointer/.max_size) over plain numpy dicts, and can be
attached to an engine-backed agent: ``attach(agent)`` uploads the arrays to HBM once (``fql_dataset_upload``), after
which ``agent.update_from_dataset(batch_size, idxs=ds.get_random_idxs(B))`` -- or no idxs at all for the engine's own
index stream -- replaces ``agent.update(ds.sample(B))`` (main.py:201,216) without any per-step H2D copy, and
``add_transition`` also writes the row into the device ring (``fql_dataset_add``).  Frame stacking / image
augmentation (utils/datasets.py:73-92,102-112) belong to the visual path and are not mirrored.
"""
from __future__ import annotations

import numpy as np

KEYS = ('observations', 'actions', 'rewards', 'masks', 'next_observations', 'terminals')


def get_size(data) -> int:
    """utils/datasets.py:11-14."""
    return max(len(v) for v in data.values())


class Dataset(dict):
    """utils/datasets.py:36-100 (a FrozenDict of arrays in the reference; a dict subclass here)."""

    @classmethod
    def create(cls, freeze=True, **fields):
        assert 'observations' in fields
        data = {k: np.asarray(v) for k, v in fields.items()}
        if freeze:
            for v in data.values():
                v.setflags(write=False)
        return cls(data)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.size = get_size(self)
        self._agent = None

    def get_random_idxs(self, num_idxs):
        """utils/datasets.py:64-66: np.random.randint(self.size, size=num_idxs) on the global legacy stream."""
        return np.random.randint(self.size, size=num_idxs)

    def get_subset(self, idxs):
        """utils/datasets.py:94-100."""
        return {k: v[idxs] for k, v in self.items()}

    def sample(self, batch_size: int, idxs=None):
        """utils/datasets.py:68-92 (state-based branch)."""
        if idxs is None:
            idxs = self.get_random_idxs(batch_size)
        return self.get_subset(idxs)

    # -- device residency ---------------------------------------------------------------------
    def attach(self, agent, capacity=None):
        """Upload the transition arrays into the agent's engine (rows [0, size))."""
        n = int(self.size)
        agent.upload_dataset({k: np.ascontiguousarray(self[k][:max(n, 1)], dtype=np.float32)[:n] if n else
                              np.zeros((0,) + self[k].shape[1:], np.float32) for k in KEYS[:5]},
                             capacity=capacity if capacity is not None else max(get_size(self), 1))
        self._agent = agent
        return self


class ReplayBuffer(Dataset):
    """utils/datasets.py:435-495."""

    @classmethod
    def create(cls, transition, size):
        buf = {k: np.zeros((size,) + np.array(v).shape, dtype=np.array(v).dtype) for k, v in transition.items()}
        return cls(buf)

    @classmethod
    def create_from_initial_dataset(cls, init_dataset, size):
        buf = {}
        for k, v in init_dataset.items():
            v = np.asarray(v)
            b = np.zeros((size,) + v.shape[1:], dtype=v.dtype)
            b[:len(v)] = v
            buf[k] = b
        ds = cls(buf)
        ds.size = ds.pointer = get_size(init_dataset)
        return ds

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.max_size = get_size(self)
        self.size = 0
        self.pointer = 0

    def add_transition(self, transition):
        """utils/datasets.py:483-491 (ring insert; size = max(pointer, size), verbatim)."""
        for k, v in self.items():
            v[self.pointer] = transition[k]
        if self._agent is not None:
            self._agent.add_transition(transition)
        self.pointer = (self.pointer + 1) % self.max_size
        self.size = max(self.pointer, self.size)

    def clear(self):
        self.size = self.pointer = 0

    def attach(self, agent, capacity=None):
        n = int(self.size)
        agent.upload_dataset({k: np.ascontiguousarray(self[k][:n], dtype=np.float32) for k in KEYS[:5]}, capacity=self.max_size)
        self._agent = agent
        return self
