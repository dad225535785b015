"""Worker of tests/test_gpu_parallel.py::test_two_processes_on_one_gpu_equal_one_step_on_the_concatenated_batch: one rank of a
world-size-2 DataParallelFQL run (gloo: RCCL refuses two ranks on one device; the wrapper's logic is backend-agnostic), both ranks on
cuda:0.  argv: rank port out_dir overlap(0|1) [B hidden]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, port, out, overlap = int(sys.argv[1]), sys.argv[2], sys.argv[3], bool(int(sys.argv[4]))
    import torch
    import torch.distributed as dist
    import fql_amd
    from fql_amd.parallel import DataParallelFQL, shard_range
    from oracle import fql_oracle as O
    from tests.util import make_problem, randomize_params
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = port
    dist.init_process_group('gloo', rank=rank, world_size=2)
    torch.cuda.set_device(0)
    od, ad = 29, 8
    B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    H = int(sys.argv[6]) if len(sys.argv) > 6 else 64
    cfg, ds, _, _ = make_problem(od, ad, B, (H, H, H, H), seed=71)
    agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)      # the SAME seed on both ranks
    if rank == 0:                                           # only rank 0 holds the "restored" state: params away from init, Adam state
        agent.set_params(randomize_params(agent.get_params(), seed=72))
        warm = O.sample_batch(ds, np.arange(B))
        agent.update(warm, noise=O.make_noise(B, ad, 73))
    dp = DataParallelFQL(agent, overlap=overlap)            # broadcasts params + Adam moments + counters from rank 0
    n = len(ds['observations'])
    lo, hi = dp.upload_shard(ds)
    assert (lo, hi) == shard_range(n, rank, 2)
    for step in range(3):
        idx_global = np.random.default_rng(100 + step).integers(0, n // 2, size=2 * B)    # per rank: B indices inside its shard
        nz = O.make_noise(2 * B, ad, 200 + step)
        mine = slice(rank * B, (rank + 1) * B)
        dp.update_from_dataset(batch_size=B, idxs=idx_global[mine], noise={k: v[mine] for k, v in nz.items()})
    torch.cuda.synchronize()
    leaves = dict(O.tree_leaves_with_path(agent.get_params()))
    nu = dict(O.tree_leaves_with_path(agent.get_opt_state()['nu']))
    np.savez(os.path.join(out, f'rank{rank}.npz'), **{k.replace('/', '|'): v for k, v in leaves.items()},
             **{'nu|' + k.replace('/', '|'): v for k, v in nu.items()})
    # engine-RNG step: ranks must draw DIFFERENT noise and rows (rank mixed into the device RNG key)
    dp.update_from_dataset(batch_size=B)
    torch.cuda.synchronize()
    import ctypes as C
    from tests.test_gpu_hardening import _workspace
    np.save(os.path.join(out, f'xbc{rank}.npy'), _workspace(agent, 1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
