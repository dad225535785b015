"""Data-parallel FQL over RCCL/xGMI: one process per GPU, replay sharded by transition index.

New functionality relative to the reference (which is single-device).  Contract
(SURVEY.md 8e): a W-rank step with per-rank batch B equals a 1-rank step on the concatenated W*B
batch, because every loss is a batch mean (agents/fql.py:37,59,66,73) -- so each rank computes the
gradient of its local mean, the flat gradient buffer is all-reduced (SUM) over RCCL, and the
optimizer kernel applies 1/W (fql_set_grad_scale) before the identical Adam/Polyak update on every
rank.  What the wrapper guarantees on top of the all-reduce:

* replicas start bit-identical: parameters, Adam moments, count and step are broadcast from rank 0;
* every rank draws DIFFERENT noise and different rows: the rank is mixed into the device RNG key
  (fql_set_rng_stream), so replicas may (and should) be created with the same seed;
* `normalize_q_loss=True` is refused for world sizes above 1: lam = 1 / mean|q| (agents/fql.py:74-76) is a
  statistic of the GLOBAL batch, and the engine computes it from the local one (a 1-float all-reduce in the
  middle of the backward pass that this wrapper does not issue) - silently wrong gradients otherwise;
* the replay is sharded PHYSICALLY: `upload_shard` sends only rows [r*N/W, (r+1)*N/W) to this rank's HBM.

The exchange itself is torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests): the
C ABI exposes the gradient buffer and its two buckets (fql_grad_buffer / fql_grad_buckets) and takes the
streams the collectives run on, so no communicator is created inside libfql_amd.so (SURVEY.md 8b listed an
`fql_allreduce_init(handle, ncclUniqueId, rank, world)`; the process group the launcher already owns makes
it redundant, and the boundary stays free of RCCL types).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_rows: int, rank: int, world: int):
    """Rank r owns rows [r*n/W, (r+1)*n/W) (contiguous; FQL samples i.i.d. single transitions,
    utils/datasets.py:64-72)."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


class _DevView:
    """__cuda_array_interface__ shim so torch can wrap the engine's gradient buffer without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


def tree_flatten(tree, prefix=''):
    out = []
    for k in sorted(tree):
        v = tree[k]
        p = f'{prefix}/{k}' if prefix else k
        out.extend(tree_flatten(v, p) if isinstance(v, dict) else [(p, v)])
    return out


def tree_unflatten(items):
    tree = {}
    for path, v in items:
        node = tree
        keys = path.split('/')
        for k in keys[:-1]:
            node = node.setdefault(k, {})
        node[keys[-1]] = v
    return tree


class DataParallelFQL:
    """Wraps an FQLAgent whose engine lives on this rank's GPU.

    The agent may also be any object with the same data-parallel surface (update_begin / update_end / grad_tensor /
    set_grad_scale / set_rng_stream / get_params / set_params / get_opt_state / set_opt_state / config) whose gradient buffer is
    a CPU torch tensor: that is how tests/test_parallel_cpu.py runs THIS class under gloo at world size 2."""

    def __init__(self, agent, process_group=None, overlap=True):
        import os
        import torch
        import torch.distributed as dist
        self.agent = agent
        self.dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        if self.world > 1 and agent.config.get('normalize_q_loss'):
            raise ValueError('normalize_q_loss=True needs mean|q| of the GLOBAL batch (agents/fql.py:74-76); the data-parallel step '
                             'computes it per rank, so it is refused for world sizes above 1')
        if hasattr(agent, 'grad_tensor'):      # CPU stand-in (tests): the gradient buffer is handed over as a tensor
            self.grads = agent.grad_tensor()
        else:
            ptr, n = agent.grad_buffer()
            self.grads = torch.as_tensor(_DevView(ptr, n), device=torch.device('cuda', torch.cuda.current_device()))
            assert self.grads.data_ptr() == ptr and self.grads.numel() == n
        self.on_gpu = self.grads.is_cuda
        agent.set_grad_scale(1.0 / self.world)
        agent.set_rng_stream(self.rank)          # same seed everywhere, different draws per rank
        self.always_reduce = bool(os.environ.get('FQL_DP_ALWAYS_REDUCE'))  # issue the collectives even at world size 1 (testing)
        # overlapped mode: the engine enqueues lane 1 (critics, BC flow) and lane 0 (Euler chain, one-step actor) on two
        # streams; the lane-1 gradient bucket (3/4 of the bytes) is all-reduced while lane 0 is still running
        self.buckets = agent.grad_buckets() if (overlap and self.on_gpu) else None
        if self.on_gpu:
            # a stream of our own: the collectives, issued on torch's current stream, must be ordered with the engine's graphs
            self.main_stream = torch.cuda.Stream()
            if self.buckets is not None:
                self.side_stream = torch.cuda.Stream()
                (o0, n0), (o1, n1) = self.buckets
                self.g0, self.g1 = self.grads[o0:o0 + n0], self.grads[o1:o1 + n1]
        self.broadcast_state()

    # -- replica state ----------------------------------------------------------------------------------------
    def _broadcast_tree(self, tree):
        import torch
        items = tree_flatten(tree)
        flat = torch.from_numpy(np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1) for _, v in items])).to(self.grads.device)
        self.dist.broadcast(flat, src=0, group=self.pg)
        flat = flat.cpu().numpy()
        out, o = [], 0
        for p, v in items:
            n = int(np.size(v))
            out.append((p, flat[o:o + n].reshape(np.shape(v))))
            o += n
        return tree_unflatten(out)

    def broadcast_state(self):
        """Replicas start from rank 0's parameters AND optimizer state (Adam mu / nu, count, step): a restored checkpoint on rank
        0 is enough.  (One-off host round trip at start-up; nothing on the step path.)"""
        import torch
        self.agent.set_params(self._broadcast_tree(self.agent.get_params()))
        opt = self.agent.get_opt_state()
        cs = torch.tensor([float(opt['count']), float(opt['step'])], dtype=torch.float64, device=self.grads.device)
        self.dist.broadcast(cs, src=0, group=self.pg)
        self.agent.set_opt_state({'mu': self._broadcast_tree(opt['mu']), 'nu': self._broadcast_tree(opt['nu']),
                                  'count': int(cs[0].item()), 'step': int(cs[1].item())})

    broadcast_params = broadcast_state   # (older name)

    def upload_shard(self, dataset, n_rows=None, **kw):
        """Physical sharding: only this rank's rows [lo, hi) of the host dataset go to its HBM.  Returns (lo, hi)."""
        n = int(n_rows if n_rows is not None else len(dataset['observations']))
        lo, hi = shard_range(n, self.rank, self.world)
        self.agent.upload_dataset({k: v[lo:hi] for k, v in dataset.items()}, **kw)
        self.shard_rows = hi - lo
        return lo, hi

    def autotune(self, batch_size=None, steps=40):
        """Time the overlapped (bucketed, two streams) and the plain (one all-reduce) step on the uploaded shard with the collectives live and
        keep the faster on every rank (the slowest rank's time decides: MAX over ranks).  These are real updates - call it during warm-up.
        Returns {'chosen', 'overlapped_us', 'plain_us'}; without the two-lane program only the plain step exists."""
        import time
        import torch
        if not self.on_gpu or self.buckets is None:
            return {'chosen': 'plain', 'overlapped_us': None, 'plain_us': None}
        saved = self.buckets
        res = {}
        for name, b in (('overlapped', saved), ('plain', None)):
            self.buckets = b
            for _ in range(5):
                self.update_from_dataset(batch_size=batch_size)
            self.dist.barrier(group=self.pg); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.update_from_dataset(batch_size=batch_size)
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / steps * 1e6], dtype=torch.float64, device=self.grads.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.pg)
            res[name] = float(t.item())
        chosen = 'overlapped' if res['overlapped'] <= res['plain'] else 'plain'
        self.buckets = saved if chosen == 'overlapped' else None
        return {'chosen': chosen, 'overlapped_us': round(res['overlapped'], 1), 'plain_us': round(res['plain'], 1)}

    # -- the step ------------------------------------------------------------------------------------------------
    def update_from_dataset(self, n_rows=None, batch_size=None, idxs=None, noise=None):
        """One synchronous data-parallel step.  After `upload_shard` the device holds this rank's rows only and indices are drawn
        over all of them; with a full copy of the dataset on every rank (`n_rows` given) they are drawn inside this rank's range."""
        if n_rows is None:
            lo, hi = 0, 0                          # the whole (already sharded) device dataset
        else:
            lo, hi = shard_range(n_rows, self.rank, self.world)
        if self.buckets is not None:
            self._overlapped(lambda s0, s1: self.agent.update_begin_split(s0, s1, idxs=idxs, shard=(lo, hi), batch_size=batch_size, noise=noise))
            return
        self._plain(lambda st: self.agent.update_begin(idxs=idxs, shard=(lo, hi), batch_size=batch_size, noise=noise, stream=st))

    def _reduce(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.pg)

    def _plain(self, begin):
        if not self.on_gpu:
            begin(None)
            if self.world > 1 or self.always_reduce:
                self._reduce(self.grads)
            self.agent.update_end(stream=None)
            return
        import torch
        outer, main = torch.cuda.current_stream(), self.main_stream
        main.wait_stream(outer)
        with torch.cuda.stream(main):
            begin(main.cuda_stream)
            if self.world > 1 or self.always_reduce:
                self._reduce(self.grads)
            self.agent.update_end(stream=main.cuda_stream)
        outer.wait_stream(main)

    def _overlapped(self, begin):
        import torch
        outer, main, side = torch.cuda.current_stream(), self.main_stream, self.side_stream
        main.wait_stream(outer)                      # earlier work of the caller first
        side.wait_stream(main)                       # lane 1 must not start before earlier work on the main stream
        begin(main.cuda_stream, side.cuda_stream)
        if self.world > 1 or self.always_reduce:
            with torch.cuda.stream(side):            # bucket 0 follows lane 1; overlaps the Euler chain on `main`
                self._reduce(self.g0)
            with torch.cuda.stream(main):
                self._reduce(self.g1)
        # Adam of the critic / BC flow right behind bucket 0 on `side` (beside lane 0's tail and bucket 1's reduce), the one-step actor's
        # and the bookkeeping on `main`, which ends behind both
        self.agent.update_end_split(main.cuda_stream, side.cuda_stream)
        outer.wait_stream(main)

    def update(self, batch, noise=None):
        if self.buckets is not None:
            self._overlapped(lambda s0, s1: self.agent.update_begin_split(s0, s1, batch=batch, noise=noise))
            return
        self._plain(lambda st: self.agent.update_begin(batch=batch, noise=noise, stream=st))

    def reduce_info(self, info):
        """Metrics across ranks at log time: means, except max/min entries."""
        import torch
        keys = list(info)
        v = torch.tensor([info[k] for k in keys], dtype=torch.float64, device=self.grads.device)
        mean = v.clone(); self.dist.all_reduce(mean, op=self.dist.ReduceOp.SUM, group=self.pg); mean /= self.world
        mx = v.clone(); self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX, group=self.pg)
        mn = v.clone(); self.dist.all_reduce(mn, op=self.dist.ReduceOp.MIN, group=self.pg)
        out = {}
        for i, k in enumerate(keys):
            out[k] = float(mx[i]) if k.endswith('max') else float(mn[i]) if k.endswith('min') else float(mean[i])
        return out
