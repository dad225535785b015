"""Data-parallel FQL over RCCL/xGMI: one process per GPU, replay sharded by transition index.

New functionality relative to the reference (which is single-device).  Contract
(SURVEY.md 8e): a W-rank step with per-rank batch B equals a 1-rank step on the concatenated W*B
batch, because every loss is a batch mean (agents/fql.py:37,59,66,73) -- so each rank computes the
gradient of its local mean, the flat gradient buffer is all-reduced (SUM) over RCCL, and the
optimizer kernel applies 1/W (fql_set_grad_scale) before the identical Adam/Polyak update on every
rank.  Params are broadcast from rank 0 at start so replicas are bit-identical.
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
    """Wraps an FQLAgent whose engine lives on this rank's GPU."""

    def __init__(self, agent, process_group=None, overlap=True):
        import torch
        import torch.distributed as dist
        self.agent = agent
        self.dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        ptr, n = agent.grad_buffer()
        self.grads = torch.as_tensor(_DevView(ptr, n), device=torch.device('cuda', torch.cuda.current_device()))
        assert self.grads.data_ptr() == ptr and self.grads.numel() == n
        agent.set_grad_scale(1.0 / self.world)
        import os
        self.always_reduce = bool(os.environ.get('FQL_DP_ALWAYS_REDUCE'))  # issue the collectives even at world size 1 (testing)
        # overlapped mode: the engine enqueues lane 1 (critics, BC flow) and lane 0 (Euler chain, one-step actor) on two
        # streams; the lane-1 gradient bucket (3/4 of the bytes) is all-reduced while lane 0 is still running
        self.buckets = agent.grad_buckets() if overlap else None
        # a stream of our own: torch's default stream has the NULL handle, which the engine would read as "use your own stream" -
        # and the collectives, issued on torch's current stream, must be ordered with the engine's graphs
        self.main_stream = torch.cuda.Stream()
        if self.buckets is not None:
            self.side_stream = torch.cuda.Stream()
            (o0, n0), (o1, n1) = self.buckets
            self.g0, self.g1 = self.grads[o0:o0 + n0], self.grads[o1:o1 + n1]
        self.broadcast_params()

    def broadcast_params(self):
        """Replicas start from rank 0's parameters and optimizer state."""
        import torch
        dev = self.grads.device
        for getter, setter in ((self.agent.get_params, self.agent.set_params),):
            items = tree_flatten(getter())
            flat = torch.from_numpy(np.concatenate([v.reshape(-1) for _, v in items])).to(dev)
            self.dist.broadcast(flat, src=0, group=self.pg)
            flat = flat.cpu().numpy()
            out, o = [], 0
            for p, v in items:
                out.append((p, flat[o:o + v.size].reshape(v.shape)))
                o += v.size
            setter(tree_unflatten(out))

    def update_from_dataset(self, n_rows, batch_size=None, idxs=None, noise=None):
        """One synchronous data-parallel step; indices are drawn from this rank's shard."""
        lo, hi = shard_range(n_rows, self.rank, self.world)
        if self.buckets is not None:
            self._overlapped(lambda s0, s1: self.agent.update_begin_split(s0, s1, idxs=idxs, shard=(lo, hi), batch_size=batch_size, noise=noise))
            return
        self._plain(lambda st: self.agent.update_begin(idxs=idxs, shard=(lo, hi), batch_size=batch_size, noise=noise, stream=st))

    def _plain(self, begin):
        import torch
        outer, main = torch.cuda.current_stream(), self.main_stream
        main.wait_stream(outer)
        with torch.cuda.stream(main):
            begin(main.cuda_stream)
            if self.world > 1 or self.always_reduce:
                self.dist.all_reduce(self.grads, op=self.dist.ReduceOp.SUM, group=self.pg)
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
                self.dist.all_reduce(self.g0, op=self.dist.ReduceOp.SUM, group=self.pg)
            with torch.cuda.stream(main):
                self.dist.all_reduce(self.g1, op=self.dist.ReduceOp.SUM, group=self.pg)
        main.wait_stream(side)
        self.agent.update_end(stream=main.cuda_stream)
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
