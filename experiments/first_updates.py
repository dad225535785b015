"""Duration of each of the first updates after create + upload (submit one, wait for it), then the steady rate: what a 20-update window after a
3-update warm-up pays.  usage: python experiments/first_updates.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import fql_amd  # noqa: E402
from fql_amd.synthetic import make_synthetic_dataset  # noqa: E402

B = 256
cfg = fql_amd.get_config()
cfg.update(alpha=10.0, batch_size=B)
ds = make_synthetic_dataset(1_000_000, 29, 8, seed=0)
a = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
a.upload_dataset(ds)
a.synchronize()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    a.update_from_dataset(B)
    a.synchronize()
    ts.append((time.perf_counter() - t0) * 1e6)
print('one at a time, us:', ' '.join(f'{t:.0f}' for t in ts))
for w in (3, 3, 50):
    for _ in range(w):
        a.update_from_dataset(B)
    a.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        a.update_from_dataset(B)
    a.synchronize()
    print(f'20 back to back after {w} warm-up: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per update')
time.sleep(2.0)
for w in (3,):
    for _ in range(w):
        a.update_from_dataset(B)
    a.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        a.update_from_dataset(B)
    a.synchronize()
    print(f'after 2 s idle, 20 back to back after {w} warm-up: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per update')
