"""Host-side mirror of the reference's FQLAgent (agents/fql.py:15-246) over the C ABI.

Same method names, keyword names and return shapes as the reference so its call sites keep working:

    agent = FQLAgent.create(seed, ex_observations, ex_actions, config)      # main.py:160
    agent, info = agent.update(batch)                                         # main.py:216
    actions = agent.sample_actions(observations=ob, temperature=1, seed=key)  # main.py:225
    loss, info = agent.total_loss(val_batch, grad_params=None)                # main.py:284
    actions = agent.compute_flow_actions(observations, noises)                # agents/fql.py:155

The reference is functional (update returns a NEW agent, callers rebind); here the engine owns the
state in HBM, update mutates it and returns ``self`` -- compatible with every reference call site.
All compute happens in libfql_amd.so (HIP, gfx950); there is no Python/torch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

import numpy as np

from . import _cabi, jax_prng
from .config import ConfigDict, get_config

INFO_KEYS = (
    'critic/critic_loss', 'critic/q_mean', 'critic/q_max', 'critic/q_min',
    'actor/actor_loss', 'actor/bc_flow_loss', 'actor/distill_loss', 'actor/q_loss',
    'actor/q', 'actor/mse', 'grad/max', 'grad/min', 'grad/norm',
)
NOISE_KEYS = ('eps1', 'x0', 't', 'z', 'eps2')
BATCH_KEYS = ('observations', 'actions', 'rewards', 'masks', 'next_observations')


def _torch():
    import torch
    return torch


def _seed_to_u64(seed) -> int:
    """Accept what reference call sites pass as `seed`: None, int, or a JAX-style uint32[2] key."""
    if seed is None:
        return 0
    a = np.asarray(seed)
    if a.ndim == 0:
        return int(a) & 0xFFFFFFFFFFFFFFFF
    a = a.astype(np.uint64).reshape(-1)
    v = 0
    for x in a[:2]:
        v = ((v << 32) | (int(x) & 0xFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    return v


class _Arg:
    """fp32 contiguous view of a user array + the pointer handed to the C ABI (device or host)."""

    def __init__(self, x, shape=None, u8=False):
        self.keep = None
        if x is None:
            self.ptr = None
            return
        if hasattr(x, 'data_ptr'):  # torch tensor (cuda or cpu)
            torch = _torch()
            t = x.detach()
            want = torch.uint8 if u8 else torch.float32
            if t.dtype != want:
                t = t.to(want)
            t = t.contiguous()
            if shape is not None and int(np.prod(shape)) != t.numel():
                raise ValueError(f'expected {int(np.prod(shape))} elements, got shape {tuple(t.shape)}')
            self.keep = t
            self.ptr = t.data_ptr()
        else:
            a = np.ascontiguousarray(np.asarray(x), dtype=np.uint8 if u8 else np.float32)
            if shape is not None and int(np.prod(shape)) != a.size:
                raise ValueError(f'expected {int(np.prod(shape))} elements, got shape {a.shape}')
            self.keep = a
            self.ptr = a.ctypes.data


class _Pending:
    """The 13 scalars of one update, fetched from the engine's pinned snapshot ring on first use (one blocking wait, then cached)."""

    def __init__(self, agent, ticket):
        self.agent, self.ticket, self.vals = agent, ticket, None

    def get(self):
        if self.vals is None:
            buf = (C.c_float * _cabi.FQL_NUM_INFO)()
            self.agent._check(self.agent._lib.fql_info_wait(self.agent._h, self.ticket, buf))
            self.vals = [float(buf[i]) for i in range(_cabi.FQL_NUM_INFO)]
            self.agent = None
        return self.vals


class LazyScalar:
    """One info value of an update that may not have run yet - what a JAX device scalar is in the reference's info dict (agents/fql.py:39-44;
    main.py:276 reads them at log time).  float(x), arithmetic, comparisons, formatting, numpy conversion all block until the update has run and
    then behave as the plain float.  Pickles as a float."""
    __slots__ = ('_p', '_i')

    def __init__(self, pending, i):
        self._p, self._i = pending, i

    def __float__(self): return self._p.get()[self._i]
    def item(self): return float(self)
    def __int__(self): return int(float(self))
    def __bool__(self): return bool(float(self))
    def __round__(self, n=None): return round(float(self), n)
    def __repr__(self): return repr(float(self))
    __str__ = __repr__
    def __format__(self, spec): return format(float(self), spec)
    def __array__(self, dtype=None, copy=None): return np.asarray(float(self), dtype=dtype)
    def __reduce__(self): return (float, (float(self),))
    def __hash__(self): return hash(float(self))
    def __eq__(self, o): return float(self) == o
    def __ne__(self, o): return float(self) != o
    def __lt__(self, o): return float(self) < o
    def __le__(self, o): return float(self) <= o
    def __gt__(self, o): return float(self) > o
    def __ge__(self, o): return float(self) >= o
    def __neg__(self): return -float(self)
    def __pos__(self): return float(self)
    def __abs__(self): return abs(float(self))
    def __add__(self, o): return float(self) + o
    def __radd__(self, o): return o + float(self)
    def __sub__(self, o): return float(self) - o
    def __rsub__(self, o): return o - float(self)
    def __mul__(self, o): return float(self) * o
    def __rmul__(self, o): return o * float(self)
    def __truediv__(self, o): return float(self) / o
    def __rtruediv__(self, o): return o / float(self)
    def __pow__(self, o): return float(self) ** o
    def __rpow__(self, o): return o ** float(self)


class LazyInfo(dict):
    """What `update` returns as `info` (the reference returns device scalars nobody waits for until main.py:276 logs them, so its training loop
    never synchronises per step): a REAL dict holding the 13 keys from the start - len(), iteration, json / pickle see all of them - whose values
    are LazyScalar objects that block on first use.  `to_dict()` gives plain floats (json.dumps(info.to_dict()), or json.dumps(info,
    default=float)); pickling and copy() give a plain dict of floats.  Holding it across more than 64 later lazy updates without reading it
    raises on first use (the engine keeps 64 snapshots)."""

    def __init__(self, agent, ticket):
        p = _Pending(agent, ticket)
        super().__init__((k, LazyScalar(p, i)) for i, k in enumerate(INFO_KEYS))

    def to_dict(self):
        return {k: float(v) for k, v in self.items()}

    def copy(self):
        return self.to_dict()

    def __reduce__(self):
        return (dict, (self.to_dict(),))


class FQLAgent:
    """Flow Q-learning agent backed by the MI355X step engine."""

    def __init__(self, handle, config, seed):
        self._lib = _cabi.load()
        self._h = handle
        self.config = config
        self._seed = seed
        self._keep = []
        self._sample_calls = 0
        # agents/fql.py:189-190: rng = PRNGKey(seed); rng, init_rng = split(rng, 2).  Only used when config['rng'] ==
        # 'jax' (noise drawn on the host with the reference's key derivation) or when a caller passes JAX keys.
        self.rng = jax_prng.split(jax_prng.PRNGKey(seed), 2, partitionable=bool(self._jax_mode()))[0]

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def create(cls, seed, ex_observations, ex_actions, config):
        """agents/fql.py:173-246.  `config`: dict-like with the keys of get_config()."""
        lib = _cabi.load()
        cfg = get_config()
        cfg.update(dict(config))
        ex_observations = np.asarray(ex_observations) if not hasattr(ex_observations, 'shape') else ex_observations
        ob_dims = tuple(int(d) for d in ex_observations.shape[1:])
        action_dim = int(ex_actions.shape[-1])
        c = _cabi.FqlConfig()
        lib.fql_default_config(C.byref(c))
        if cfg.get('encoder') is not None:
            # agents/fql.py:196-202: one encoder per module in front of the MLPs; observations are uint8 images [H, W, C]
            if cfg['encoder'] not in ('impala_small', 'impala'):
                raise NotImplementedError(f"encoder {cfg['encoder']!r}: 'impala_small' and 'impala' (utils/encoders.py:104,106) are built")
            if len(ob_dims) != 3:
                raise ValueError(f'image observations [H, W, C] expected with an encoder, got ob_dims={ob_dims}')
            c.encoder, c.img_h, c.img_w, c.img_c = (1 if cfg['encoder'] == 'impala_small' else 2), ob_dims[0], ob_dims[1], ob_dims[2]
            c.obs_dim = 1
        else:
            if len(ob_dims) != 1:
                raise ValueError(f'state-based observations expected, got ob_dims={ob_dims}')
            c.obs_dim = int(ob_dims[0])
        c.act_dim = action_dim
        ah, vh = tuple(cfg['actor_hidden_dims']), tuple(cfg['value_hidden_dims'])
        if len(ah) > _cabi.FQL_MAX_HIDDEN or len(vh) > _cabi.FQL_MAX_HIDDEN:
            raise ValueError('too many hidden layers')
        c.num_actor_hidden, c.num_value_hidden = len(ah), len(vh)
        for i, v in enumerate(ah):
            c.actor_hidden[i] = int(v)
        for i, v in enumerate(vh):
            c.value_hidden[i] = int(v)
        c.layer_norm, c.actor_layer_norm = int(bool(cfg['layer_norm'])), int(bool(cfg['actor_layer_norm']))
        c.lr, c.discount, c.tau, c.alpha = float(cfg['lr']), float(cfg['discount']), float(cfg['tau']), float(cfg['alpha'])
        if cfg['q_agg'] not in ('mean', 'min'):
            raise ValueError(f"q_agg must be 'mean' or 'min', got {cfg['q_agg']!r}")
        c.q_agg = 1 if cfg['q_agg'] == 'min' else 0
        c.flow_steps = int(cfg['flow_steps'])
        c.normalize_q_loss = int(bool(cfg['normalize_q_loss']))
        c.batch_size = int(cfg['batch_size'])
        prec = cfg.get('precision', 'fp32')
        if prec not in ('fp32', 'bf16x3', 0, 2):
            raise ValueError(f"precision must be 'fp32' or 'bf16x3', got {prec!r}")
        c.precision = 2 if prec in ('bf16x3', 2) else 0
        h = C.c_void_p()
        rc = lib.fql_create(C.byref(c), int(seed) & 0xFFFFFFFFFFFFFFFF, C.byref(h))
        _cabi.check(lib, None, rc)
        cfg['ob_dims'] = ob_dims
        cfg['action_dim'] = action_dim
        return cls(h, cfg, int(seed))

    def close(self):
        if getattr(self, '_h', None):
            self._lib.fql_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ----------------------------------------------------------------------------
    def _check(self, rc):
        _cabi.check(self._lib, self._h, rc)

    @staticmethod
    def _stream(args):
        """Run on torch's current stream when any argument lives on the GPU (keeps torch's caching allocator and the engine
        stream-ordered, including the dtype / contiguity temporaries made just before the call); otherwise the engine's own
        stream (NULL).  torch's DEFAULT stream has the handle 0, which the C ABI reads as "no stream": it is passed as
        FQL_STREAM_LEGACY (hipStreamLegacy) instead, so the engine's graph is enqueued on that very stream."""
        for a in args:
            k = getattr(a, 'keep', None)
            if k is not None and hasattr(k, 'is_cuda') and k.is_cuda:
                h = _torch().cuda.current_stream().cuda_stream
                return h if h else _cabi.FQL_STREAM_LEGACY
        return None

    def _batch_args(self, batch):
        B = int(np.shape(batch['actions'])[0]) if not hasattr(batch['actions'], 'shape') else int(batch['actions'].shape[0])
        obd, ad = tuple(self.config['ob_dims']), self.config['action_dim']
        shapes = {'observations': (B,) + obd, 'actions': (B, ad), 'rewards': (B,), 'masks': (B,), 'next_observations': (B,) + obd}
        for k in BATCH_KEYS:
            if k not in batch:
                raise KeyError(f'batch is missing {k!r}')
        vis = len(obd) == 3   # image observations travel as uint8 (utils/encoders.py:84 divides by 255 on the device)
        return B, [_Arg(batch[k], shapes[k], u8=vis and k.endswith('observations')) for k in BATCH_KEYS]

    def _noise_args(self, noise, B):
        if noise is None:
            return None, []
        ad = self.config['action_dim']
        shapes = {'eps1': (B, ad), 'x0': (B, ad), 't': (B,), 'z': (B, ad), 'eps2': (B, ad)}
        args = [_Arg(noise.get(k), shapes[k]) for k in NOISE_KEYS]
        nz = _cabi.FqlNoise(*[a.ptr for a in args])
        return nz, args

    def _ensure_batch(self, B):
        if B != self.config['batch_size']:
            self._check(self._lib.fql_set_batch_size(self._h, B))
            self.config['batch_size'] = B

    # -- reference API ----------------------------------------------------------------------
    def update(self, batch, noise: Optional[Dict[str, Any]] = None, want_info=True):
        """agents/fql.py:122-133.  Returns (agent, info).  `noise` (optional) supplies the five random
        tensors explicitly (parity runs); by default they come from the engine's device RNG.
        `info` is a LazyInfo: a dict of the 13 scalars that is read from the device on first access, so a loop that only logs
        every N steps (main.py:216,276) never synchronises in between.  want_info=False returns None (no snapshot at all)."""
        B, args = self._batch_args(batch)
        self._ensure_batch(B)
        jax_mode = self._jax_mode()
        if noise is None and jax_mode is not None and not self.config.get('rng_device', True):
            self.rng, noise = jax_prng.fql_update_noise(self.rng, B, self.config['action_dim'], partitionable=jax_mode)
        nz, nargs = self._noise_args(noise, B)
        stream = self._stream(args + nargs)
        if noise is None and jax_mode is not None:
            # the reference's key derivation on the host (a dozen 2-word hashes), the five tensors on the device: nothing but 10 words
            # of keys crosses PCIe (config['rng_device']=False: tensors drawn on the host by fql_amd/jax_prng.py instead)
            self.rng, keys = jax_prng.fql_update_keys(self.rng, partitionable=jax_mode)
            nz = self._device_noise(keys, jax_mode, B, stream)
        self._check(self._lib.fql_update(self._h, *[a.ptr for a in args], B, C.byref(nz) if nz else None, None, stream))
        self._keep = (args, nargs)
        return self, (self._lazy_info(stream) if want_info else None)

    def _jax_mode(self):
        """None: engine RNG; False / True: JAX key derivation with the original / partitionable threefry layout."""
        r = self.config.get('rng')
        if r == 'jax':
            return False
        if r == 'jax_partitionable':
            return True
        return None

    def _device_noise(self, keys, partitionable, B, stream):
        k = np.ascontiguousarray(np.stack([np.asarray(keys[n], dtype=np.uint32).reshape(2) for n in NOISE_KEYS]))
        nz = _cabi.FqlNoise()
        self._check(self._lib.fql_noise_from_jax_keys(self._h, k.ctypes.data, int(bool(partitionable)), B, C.byref(nz), stream))
        return nz

    def _lazy_info(self, stream):
        t = C.c_uint64()
        self._check(self._lib.fql_info_enqueue(self._h, stream, C.byref(t)))
        return LazyInfo(self, int(t.value))

    def _resident_noise(self, noise, B, stream):
        """(fql_noise or None, keep-alive) for the device-resident update paths: explicit tensors, else - with config['rng'] = 'jax' /
        'jax_partitionable' - the reference's key derivation advanced on the host and the five tensors generated on the device from the derived
        keys (exactly what update() does), else None = the engine's Philox stream."""
        nz, nargs = self._noise_args(noise, B)
        jax_mode = self._jax_mode()
        if noise is None and jax_mode is not None:
            self.rng, keys = jax_prng.fql_update_keys(self.rng, partitionable=jax_mode)
            nz = self._device_noise(keys, jax_mode, B, stream)
        return nz, nargs

    def _host_idx(self, idxs, n, rows, what):
        """int64 view of HOST gather indices, checked against the rows the device array holds: an index >= size would train on zero rows or read
        past the array on the device.  Device tensors are passed through unchecked (documented in INTEGRATION.md)."""
        if hasattr(idxs, 'data_ptr'):
            keep = idxs.long().contiguous()
            if keep.numel() != n:
                raise ValueError(f'{what} must have {n} entries')
            if not keep.is_cuda:
                lo, hi = (int(keep.min()), int(keep.max())) if n else (0, -1)
                if lo < 0 or hi >= rows:
                    raise ValueError(f'{what} out of range: [{lo}, {hi}] for {rows} rows')
            return keep, keep.data_ptr()
        keep = np.ascontiguousarray(idxs, dtype=np.int64)
        if keep.size != n:
            raise ValueError(f'{what} must have {n} entries')
        if n and (keep.min() < 0 or keep.max() >= rows):
            raise ValueError(f'{what} out of range: [{int(keep.min())}, {int(keep.max())}] for {rows} rows')
        return keep, keep.ctypes.data

    def read_info(self) -> Dict[str, float]:
        """The 13 info scalars of the last update (blocks until that update has run)."""
        buf = (C.c_float * _cabi.FQL_NUM_INFO)()
        self._check(self._lib.fql_read_info(self._h, buf))
        return {k: float(buf[i]) for i, k in enumerate(INFO_KEYS)}

    def synchronize(self) -> str:
        """Host wait for every update enqueued so far (jax.block_until_ready(agent)): the engine's HIP stream and the hardware queues of
        its own that stream-less updates run on, which torch.cuda.synchronize() does not see.  Returns where the last update ran:
        'aql' (the engine's own queues) or 'graph' (a captured graph on a HIP stream)."""
        m = C.c_int(0)
        self._check(self._lib.fql_synchronize(self._h, C.byref(m)))
        return 'aql' if m.value else 'graph'

    def total_loss(self, batch, grad_params=None, rng=None, noise: Optional[Dict[str, Any]] = None):
        """agents/fql.py:94-111 with grad_params=None (the validation probe, main.py:284)."""
        if grad_params is not None:
            raise NotImplementedError('total_loss with traced grad_params is jax.grad plumbing; use update()')
        B, args = self._batch_args(batch)
        self._ensure_batch(B)
        if noise is None and (rng is not None or self._jax_mode() is not None):
            noise = jax_prng.fql_total_loss_noise(self.rng if rng is None else rng, B, self.config['action_dim'],
                                                  partitionable=bool(self._jax_mode()))
        nz, nargs = self._noise_args(noise, B)
        loss = C.c_float()
        info = (C.c_float * 10)()
        stream = self._stream(args + nargs)
        self._check(self._lib.fql_total_loss(self._h, *[a.ptr for a in args], B, C.byref(nz) if nz else None,
                                             C.byref(loss), info, stream))
        return float(loss.value), {k: float(info[i]) for i, k in enumerate(INFO_KEYS[:10])}

    def sample_actions(self, observations, seed=None, temperature=1.0, noises=None):
        """agents/fql.py:135-153: clip(onestep(obs, N(0, I))).  `temperature` is accepted and ignored,
        as in the reference.  Leading dims of `observations` are preserved ([od] -> [ad])."""
        return self._eval(observations, noises, seed, flow=False)

    def compute_flow_actions(self, observations, noises):
        """agents/fql.py:155-171."""
        return self._eval(observations, noises, None, flow=True)

    def _eval(self, observations, noises, seed, flow):
        obd, ad = tuple(self.config['ob_dims']), self.config['action_dim']
        vis = len(obd) == 3
        od = obd[0] if not vis else None
        is_torch = hasattr(observations, 'data_ptr')
        if vis and tuple(observations.shape[-3:]) != obd:
            raise ValueError(f'observations must end in {obd}, got {tuple(observations.shape)}')
        lead = tuple(observations.shape[:-3]) if vis else tuple(observations.shape[:-1])
        if not flow and noises is None and seed is not None and np.ndim(seed) == 1 and np.size(seed) == 2:
            # a JAX key (what main.py:225 / utils/evaluation.py:98-101 pass): draw exactly the noise the reference draws
            noises = jax_prng.sample_actions_noise(np.asarray(seed), lead, ad, partitionable=bool(self._jax_mode()))
        if not vis and int(observations.shape[-1]) != od:
            raise ValueError(f'observations last dim must be {od}, got {tuple(observations.shape)}')
        n = int(np.prod(lead)) if lead else 1
        o = _Arg(observations, (n,) + obd, u8=vis)
        z = _Arg(noises, (n, ad))
        if is_torch and observations.is_cuda:
            torch = _torch()
            out = torch.empty(lead + (ad,), dtype=torch.float32, device=observations.device)
            optr = out.data_ptr()
        else:
            out = np.empty(lead + (ad,), dtype=np.float32)
            optr = out.ctypes.data
        stream = self._stream([o, z])
        if flow:
            self._check(self._lib.fql_flow_actions(self._h, o.ptr, z.ptr, n, optr, stream))
        else:
            if seed is None and noises is None:
                self._sample_calls += 1
                sd = (self._seed << 20) ^ self._sample_calls
            else:
                sd = _seed_to_u64(seed)
            self._check(self._lib.fql_sample_actions(self._h, o.ptr, n, z.ptr, sd & 0xFFFFFFFFFFFFFFFF, optr, stream))
        return out

    # -- device-resident dataset (utils/datasets.py Dataset/ReplayBuffer on the GPU) -------------
    def upload_dataset(self, dataset, capacity: Optional[int] = None, frame_stack: Optional[int] = None, p_aug: Optional[float] = None):
        """Dataset.create(...) arrays -> HBM.  Visual agents: `observations` / `next_observations` are the uint8 FRAMES
        [N, H, W, C / frame_stack] and `terminals` marks episode ends; frame stacking (main.py:120-121 sets
        dataset.frame_stack) and the random crop (dataset.p_aug) then happen in the device-side gather."""
        if len(self.config['ob_dims']) == 3:
            fs = int(frame_stack if frame_stack is not None else (self.config.get('frame_stack') or 1))
            pa = float(p_aug if p_aug is not None else (self.config.get('p_aug') or 0.0))
            n = int(len(dataset['observations']))
            H, W, Cc = self.config['ob_dims']
            fshape = (n, H, W, Cc // fs)
            fr = _Arg(dataset['observations'], fshape, u8=True)
            nf = _Arg(dataset['next_observations'], fshape, u8=True)
            rest = [_Arg(dataset[k]) for k in ('actions', 'rewards', 'masks', 'terminals')]
            self._check(self._lib.fql_dataset_upload_frames(self._h, n, fr.ptr, nf.ptr, *[a.ptr for a in rest], fs, pa))
            self._frame_stack = fs     # (remembered for the shape check of later ring inserts)
            return
        n = int(len(dataset['observations']))
        cap = int(capacity) if capacity is not None else max(n, 1)
        args = [_Arg(dataset[k]) for k in BATCH_KEYS]
        self._check(self._lib.fql_dataset_upload(self._h, n, cap, *[a.ptr for a in args]))

    def reserve_dataset(self, capacity: int):
        """ReplayBuffer.create_from_initial_dataset(dataset, size) (utils/datasets.py:457-473, main.py:111-115): the uploaded dataset
        becomes a ring of `capacity` rows (state or uint8 frames); `add_transition` then inserts at the pointer."""
        self._check(self._lib.fql_dataset_reserve(self._h, int(capacity)))

    def create_replay_buffer(self, size: int):
        """ReplayBuffer.create(example_transition, size) (utils/datasets.py:441-455; main.py:106-109): an empty device ring beside the
        training dataset, the second source of `update_balanced`."""
        self._check(self._lib.fql_replay_create(self._h, int(size)))

    def add_transition(self, transition, replay: bool = False):
        """ReplayBuffer.add_transition (utils/datasets.py:483-491) into the dataset ring (replay=False: the dataset IS the replay
        buffer, main.py:111-115) or into the separate replay ring (replay=True, balanced sampling).  Visual agents pass single uint8
        frames [H, W, C / frame_stack] as observations / next_observations."""
        a = _Arg(transition['actions'], (self.config['action_dim'],))
        r, m = float(transition['rewards']), float(transition['masks'])
        if len(self.config['ob_dims']) == 3:
            H, W, Cc = self.config['ob_dims']
            fs = int(getattr(self, '_frame_stack', 0) or 0)
            fr = _Arg(transition['observations'], u8=True); nf = _Arg(transition['next_observations'], u8=True)
            nel = fr.keep.numel() if hasattr(fr.keep, 'numel') else fr.keep.size
            if fs and nel != H * W * (Cc // fs):
                raise ValueError(f'expected one uint8 frame of {H}x{W}x{Cc // fs}, got {nel} elements')
            f = self._lib.fql_replay_add_frames if replay else self._lib.fql_dataset_add_frames
            self._check(f(self._h, fr.ptr, nf.ptr, a.ptr, r, m))
            return
        od = self.config['ob_dims'][0]
        o = _Arg(transition['observations'], (od,)); no = _Arg(transition['next_observations'], (od,))   # (raises on a wrong element count)
        f = self._lib.fql_replay_add if replay else self._lib.fql_dataset_add
        self._check(f(self._h, o.ptr, a.ptr, r, m, no.ptr))

    def replay_size(self):
        s, p = C.c_int64(), C.c_int64()
        self._check(self._lib.fql_replay_size(self._h, C.byref(s), C.byref(p)))
        return int(s.value), int(p.value)

    def update_balanced(self, batch_size=None, idxs=None, noise=None, want_info=False, stream=None, crop_froms=None):
        """main.py:255-259 + :216 on the device: concat(train_dataset.sample(B // 2), replay_buffer.sample(B // 2)) -> agent.update.
        `idxs` = (dataset_idxs, replay_idxs), B // 2 each, or None (engine RNG)."""
        B = int(batch_size or self.config['batch_size'])
        self._ensure_batch(B)
        nz, nargs = self._resident_noise(noise, B, stream)
        ia = ib = None
        keep = None
        if idxs is not None:
            ka, ia = self._host_idx(idxs[0], B // 2, self.dataset_size()[0], 'dataset idxs')
            kb, ib = self._host_idx(idxs[1], B // 2, self.replay_size()[0], 'replay idxs')
            keep = (ka, kb)
        cp = None
        if crop_froms is not None:
            ck = np.ascontiguousarray(crop_froms, dtype=np.int32)
            if ck.shape != (B, 2):
                raise ValueError('crop_froms must be [batch_size, 2]')
            cp, keep = ck.ctypes.data, (keep, ck)
        self._check(self._lib.fql_update_balanced(self._h, ia, ib, cp, B, C.byref(nz) if nz else None, None, stream))
        self._keep = (keep, nargs)
        return self, (self._lazy_info(stream) if want_info else None)

    def dataset_size(self):
        s, p = C.c_int64(), C.c_int64()
        self._check(self._lib.fql_dataset_size(self._h, C.byref(s), C.byref(p)))
        return int(s.value), int(p.value)

    def update_from_dataset(self, batch_size=None, idxs=None, noise=None, shard=(0, 0), want_info=False, stream=None,
                            crop_froms=None):
        """train_dataset.sample(B) + agent.update(batch) (main.py:201,216) without leaving the device.  Visual agents:
        `crop_froms` int [B, 2] fixes the random-crop offsets Dataset.augment would draw (None: engine RNG with p_aug)."""
        B = int(batch_size or self.config['batch_size'])
        self._ensure_batch(B)
        nz, nargs = self._resident_noise(noise, B, stream)
        ip, keep = None, None
        if idxs is not None:
            keep, ip = self._host_idx(idxs, B, self.dataset_size()[0], 'idxs')
        if crop_froms is not None:
            ck = np.ascontiguousarray(crop_froms, dtype=np.int32)
            if ck.shape != (B, 2):
                raise ValueError('crop_froms must be [batch_size, 2]')
            self._check(self._lib.fql_update_from_frames(self._h, ip, ck.ctypes.data, B, int(shard[0]), int(shard[1]),
                                                         C.byref(nz) if nz else None, None, stream))
            keep = (keep, ck)
        else:
            self._check(self._lib.fql_update_from_dataset(self._h, ip, B, int(shard[0]), int(shard[1]),
                                                          C.byref(nz) if nz else None, None, stream))
        self._keep = (keep, nargs)
        return self, (self._lazy_info(stream) if want_info else None)

    # -- data-parallel halves (fql_amd/parallel.py drives these) ---------------------------------
    def update_begin(self, batch=None, noise=None, idxs=None, shard=(0, 0), batch_size=None, stream=None):
        if batch is not None:
            B, args = self._batch_args(batch)
            self._ensure_batch(B)
            st = stream if stream is not None else self._stream(args)
            nz, nargs = self._resident_noise(noise, B, st)
            self._check(self._lib.fql_update_begin(self._h, *[a.ptr for a in args], B, C.byref(nz) if nz else None, st))
            self._keep = (args, nargs)
        else:
            B = int(batch_size or self.config['batch_size'])
            self._ensure_batch(B)
            nz, nargs = self._resident_noise(noise, B, stream)
            ip, keep = None, None
            if idxs is not None:
                keep, ip = self._host_idx(idxs, B, self.dataset_size()[0], 'idxs')
            self._check(self._lib.fql_update_from_dataset_begin(self._h, ip, B, int(shard[0]), int(shard[1]),
                                                                C.byref(nz) if nz else None, stream))
            self._keep = (keep, nargs)

    def update_end(self, stream=None):
        self._check(self._lib.fql_update_end(self._h, None, stream))

    def update_end_split(self, stream0, stream1):
        """fql_update_end_split: Adam of gradient bucket 0 (critic, BC flow) on stream1, of bucket 1 (one-step actor) on stream0."""
        self._check(self._lib.fql_update_end_split(self._h, None, stream0, stream1))

    def grad_buckets(self):
        """[(offset, length)] x 2 in floats inside grad_buffer(), or None if the engine has no split program."""
        off, ln = (C.c_size_t * 2)(), (C.c_size_t * 2)()
        rc = self._lib.fql_grad_buckets(self._h, off, ln)
        if rc != 0:
            return None
        return [(int(off[0]), int(ln[0])), (int(off[1]), int(ln[1]))]

    def update_begin_split(self, stream0, stream1, batch=None, noise=None, idxs=None, shard=(0, 0), batch_size=None):
        """fql_update_begin_split / fql_update_from_dataset_begin_split: lane 0 on stream0, lane 1 on stream1."""
        if batch is not None:
            B, args = self._batch_args(batch)
            self._ensure_batch(B)
            nz, nargs = self._resident_noise(noise, B, stream0)
            self._check(self._lib.fql_update_begin_split(self._h, *[a.ptr for a in args], B, C.byref(nz) if nz else None,
                                                         stream0, stream1))
            self._keep = (args, nargs)
        else:
            B = int(batch_size or self.config['batch_size'])
            self._ensure_batch(B)
            nz, nargs = self._resident_noise(noise, B, stream0)
            ip, keep = None, None
            if idxs is not None:
                keep, ip = self._host_idx(idxs, B, self.dataset_size()[0], 'idxs')
            self._check(self._lib.fql_update_from_dataset_begin_split(self._h, ip, B, int(shard[0]), int(shard[1]),
                                                                      C.byref(nz) if nz else None, stream0, stream1))
            self._keep = (keep, nargs)

    def grad_buffer(self):
        """(device pointer, number of floats) of the flat trainable-gradient buffer."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.fql_grad_buffer(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def set_grad_scale(self, scale: float):
        self._check(self._lib.fql_set_grad_scale(self._h, float(scale)))

    def set_rng_stream(self, stream_id: int):
        """Mix `stream_id` (the rank) into the device RNG key: same seed on every replica, different draws."""
        self._check(self._lib.fql_set_rng_stream(self._h, int(stream_id) & 0xFFFFFFFFFFFFFFFF))

    # -- parameters / optimizer state (reference tree layout, utils/flax_utils.py:16-50) ---------
    def leaves(self):
        out = []
        name = C.create_string_buffer(256)
        nd = C.c_int()
        shp = (C.c_int64 * 4)()
        for i in range(self._lib.fql_num_leaves(self._h)):
            self._check(self._lib.fql_leaf_info(self._h, i, name, 256, C.byref(nd), shp))
            out.append((name.value.decode(), tuple(int(shp[k]) for k in range(nd.value))))
        return out

    def _get_tree(self, getter):
        tree: Dict[str, Any] = {}
        for path, shape in self.leaves():
            a = np.empty(shape, dtype=np.float32)
            self._check(getter(path.encode(), a.ctypes.data, a.size))
            node = tree
            keys = path.split('/')
            for k in keys[:-1]:
                node = node.setdefault(k, {})
            node[keys[-1]] = a
        return tree

    def _set_tree(self, tree, setter, prefix=''):
        for k, v in tree.items():
            p = f'{prefix}/{k}' if prefix else k
            if isinstance(v, dict):
                self._set_tree(v, setter, p)
            else:
                a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
                self._check(setter(p.encode(), a.ctypes.data, a.size))

    def get_params(self):
        """agent.network.params as a nested dict of numpy arrays (reference leaf names and shapes)."""
        return self._get_tree(lambda p, d, n: self._lib.fql_get_param(self._h, p, d, n))

    def set_params(self, params):
        self._set_tree(params, lambda p, d, n: self._lib.fql_set_param(self._h, p, d, n))

    def get_opt_state(self):
        """optax.adam state: {'count', 'mu', 'nu'}."""
        c, s = C.c_int64(), C.c_int64()
        self._check(self._lib.fql_get_step(self._h, C.byref(c), C.byref(s)))
        return {'count': int(c.value), 'step': int(s.value),
                'mu': self._get_tree(lambda p, d, n: self._lib.fql_get_opt_state(self._h, 0, p, d, n)),
                'nu': self._get_tree(lambda p, d, n: self._lib.fql_get_opt_state(self._h, 1, p, d, n))}

    def set_opt_state(self, state):
        self._set_tree(state['mu'], lambda p, d, n: self._lib.fql_set_opt_state(self._h, 0, p, d, n))
        self._set_tree(state['nu'], lambda p, d, n: self._lib.fql_set_opt_state(self._h, 1, p, d, n))
        self._check(self._lib.fql_set_step(self._h, int(state['count']), int(state.get('step', state['count'] + 1))))

    def stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.fql_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {'launches_per_update': int(a.value), 'macs_per_update': int(b.value), 'param_count': int(c.value)}
