"""Host enqueue time vs total time of the state-config update loop (is the 470 us/step host-bound?)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import fql_amd
from oracle import fql_oracle as O
B = 256
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=B)
ds = O.make_synthetic_dataset(100000, 29, 8, seed=0)
torch.zeros(1, device='cuda')
agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
agent.upload_dataset(ds)
st = torch.cuda.current_stream().cuda_stream
for _ in range(300): agent.update_from_dataset(B, stream=st)
torch.cuda.synchronize()
for n in (200, 2000):
    t0 = time.perf_counter()
    for _ in range(n): agent.update_from_dataset(B, stream=st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('n=%d enqueue %.1f us/step, total %.1f us/step' % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
