#!/usr/bin/env python3
"""Update rate with or without torch in the process (torch bundles its own HIP runtime; without it the system ROCm runtime is used).
usage: rate.py [torch|notorch] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch  # noqa: F401
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=256)
ds = make_synthetic_dataset(100_000, 29, 8, seed=0)
agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
agent.upload_dataset(ds)
for _ in range(300): agent.update_from_dataset(256)
agent.read_info()
t0 = time.perf_counter()
for _ in range(N): agent.update_from_dataset(256)
agent.read_info()
dt = time.perf_counter() - t0
print(sys.argv[1] if len(sys.argv) > 1 else 'notorch', f'{N / dt:.1f} updates/s  {1e6 * dt / N:.1f} us per update')
