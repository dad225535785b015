"""Host mirror of the reference's batch source for FQL (utils/datasets.py:17-112,435-495).

``Dataset`` / ``ReplayBuffer`` keep the reference's API (create, sample, get_random_idxs, get_subset,
create_from_initial_dataset, add_transition, clear, .size/.pointer/.max_size) over plain numpy dicts, and can be
attached to an engine-backed agent: ``attach(agent)`` uploads the arrays to HBM once (``fql_dataset_upload``), after
which ``agent.update_from_dataset(batch_size, idxs=ds.get_random_idxs(B))`` -- or no idxs at all for the engine's own
index stream -- replaces ``agent.update(ds.sample(B))`` (main.py:201,216) without any per-step H2D copy, and
``add_transition`` also writes the row into the device ring (``fql_dataset_add``).  Image datasets: ``frame_stack`` and
``p_aug`` are attributes set from outside exactly as main.py:116-121 does; ``sample`` then stacks frames (clamped to the episode
start) and applies the edge-padded random crop on the host (utils/datasets.py:73-112), and ``attach`` uploads the uint8 frames
so that the same happens inside the device gather (``fql_dataset_upload_frames``).
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
        self.frame_stack = None   # number of frames to stack; set outside the class (utils/datasets.py:54, main.py:120)
        self.p_aug = None         # image augmentation probability; set outside the class (utils/datasets.py:55, main.py:121)
        if 'terminals' in self:   # utils/datasets.py:58-62
            self.terminal_locs = np.nonzero(np.asarray(self['terminals']) > 0)[0]
            self.initial_locs = np.concatenate([[0], self.terminal_locs[:-1] + 1]).astype(np.int64)

    def get_random_idxs(self, num_idxs):
        """utils/datasets.py:64-66: np.random.randint(self.size, size=num_idxs) on the global legacy stream."""
        return np.random.randint(self.size, size=num_idxs)

    def get_subset(self, idxs):
        """utils/datasets.py:94-100."""
        return {k: v[idxs] for k, v in self.items()}

    def sample(self, batch_size: int, idxs=None):
        """utils/datasets.py:68-92."""
        if idxs is None:
            idxs = self.get_random_idxs(batch_size)
        batch = self.get_subset(idxs)
        if self.frame_stack is not None:
            idxs = np.asarray(idxs)
            init = self.initial_locs[np.searchsorted(self.initial_locs, idxs, side='right') - 1]
            obs, next_obs = [], []   # [ob[t - k + 1], ..., ob[t]] and [ob[t - k + 2], ..., ob[t], next_ob[t]]
            for i in reversed(range(self.frame_stack)):
                cur = np.maximum(idxs - i, init)   # the episode's first frame when the index runs out of it
                obs.append(self['observations'][cur])
                if i != self.frame_stack - 1:
                    next_obs.append(self['observations'][cur])
            next_obs.append(self['next_observations'][idxs])
            batch['observations'] = np.concatenate(obs, axis=-1)
            batch['next_observations'] = np.concatenate(next_obs, axis=-1)
        if self.p_aug is not None and np.random.rand() < self.p_aug:
            self.augment(batch, ['observations', 'next_observations'])
        return batch

    def augment(self, batch, keys, crop_froms=None):
        """utils/datasets.py:102-112 + random_crop :17-33: pad 3 with edge values, slice [H, W] at per-sample offsets that are
        shared by all `keys`."""
        padding = 3
        n = len(batch[keys[0]])
        if crop_froms is None:
            crop_froms = np.random.randint(0, 2 * padding + 1, (n, 2))
        for key in keys:
            arr = batch[key]
            if arr.ndim != 4:
                continue
            h, w = arr.shape[1:3]
            padded = np.pad(arr, ((0, 0), (padding, padding), (padding, padding), (0, 0)), mode='edge')
            out = np.empty_like(arr)
            for b in range(n):
                y, x = int(crop_froms[b][0]), int(crop_froms[b][1])
                out[b] = padded[b, y:y + h, x:x + w]
            batch[key] = out
        return crop_froms

    # -- device residency ---------------------------------------------------------------------
    def attach(self, agent, capacity=None):
        """Upload the transition arrays into the agent's engine (rows [0, size))."""
        n = int(self.size)
        if np.asarray(self['observations']).ndim == 4:   # image frames: stacking / crop happen in the device gather
            agent.upload_dataset({k: self[k][:n] for k in KEYS}, frame_stack=self.frame_stack or 1, p_aug=self.p_aug or 0.0)
            self._agent = agent
            return self
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
        """utils/datasets.py:483-491 (ring insert; size = max(pointer, size), verbatim); mirrored into the device ring when attached."""
        for k, v in self.items():
            v[self.pointer] = transition[k]
        if self._agent is not None:
            self._agent.add_transition(transition, replay=self._as_replay)
        self.pointer = (self.pointer + 1) % self.max_size
        self.size = max(self.pointer, self.size)

    def clear(self):
        self.size = self.pointer = 0

    _as_replay = False

    def attach(self, agent, capacity=None, replay=False):
        """replay=False: this buffer IS the training dataset (main.py:111-115, create_from_initial_dataset): upload rows [0, size) into a
        device ring of max_size rows.  replay=True: the separate, initially empty buffer of balanced sampling (main.py:106-109) -> the
        engine's replay ring (agent.update_balanced draws half of each batch from it); rows already held are inserted one by one."""
        n = int(self.size)
        if replay:
            agent.create_replay_buffer(self.max_size)
            self._agent, self._as_replay = agent, True
            for i in range(n):
                agent.add_transition({k: self[k][i] for k in self}, replay=True)
            return self
        if np.asarray(self['observations']).ndim == 4:   # uint8 frames
            agent.upload_dataset({k: self[k][:n] for k in KEYS}, frame_stack=self.frame_stack or 1, p_aug=self.p_aug or 0.0)
            agent.reserve_dataset(self.max_size)
        else:
            agent.upload_dataset({k: np.ascontiguousarray(self[k][:n], dtype=np.float32) for k in KEYS[:5]}, capacity=self.max_size)
        self._agent = agent
        return self
