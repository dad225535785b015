#!/usr/bin/env python3
"""Where a 20-update timed window (the driver's bench regime: --steps 20 --warmup 5) spends its time: enqueue vs the final fence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=256)
ds = make_synthetic_dataset(100_000, 29, 8, seed=0)
agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
agent.upload_dataset(ds)
for _ in range(5): agent.update_from_dataset(256)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    ts = []
    for _ in range(20):
        agent.update_from_dataset(256); ts.append(time.perf_counter())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'enqueue {1e6*(t1-t0):.0f} us (first call {1e6*(ts[0]-t0):.0f}, later {1e6*(ts[-1]-ts[0])/19:.0f} each), fence {1e6*(t2-t1):.0f} us, total {1e6*(t2-t0):.0f} = {1e6*(t2-t0)/20:.1f} per update')
