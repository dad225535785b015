#!/usr/bin/env python3
"""Is the host ahead of the GPU?  Times the enqueue loop of N device-resident updates against the fenced total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset
od, ad, B, N = 29, 8, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=B)
ds = make_synthetic_dataset(100_000, od, ad, seed=0)
agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
agent.upload_dataset(ds)
for _ in range(300): agent.update_from_dataset(B)
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(N):
    agent.update_from_dataset(B)
    if i in (0, 9, 99, 499, 999, N - 1): marks.append((i + 1, time.perf_counter() - t0))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue loop {1e6*(t1-t0)/N:.1f} us per update; fenced {1e6*(t2-t0)/N:.1f} us per update')
print('host time after k calls (us per call so far):', [(k, round(1e6 * t / k, 1)) for k, t in marks])
