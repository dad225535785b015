"""Host enqueue cost per update against the fenced time per update, per precision: is the bf16x3 update (328 us) host-bound?
Short bursts from an idle device (no back-pressure from a full queue): enqueue us / update = what hipGraphLaunch of the update graph costs the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset
B = 256
ds = make_synthetic_dataset(100000, 29, 8, seed=0)
torch.zeros(1, device='cuda')
for prec in ('fp32', 'bf16x3'):
    cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=B, precision=prec)
    agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
    agent.upload_dataset(ds)
    for _ in range(300): agent.update_from_dataset(B)
    torch.cuda.synchronize(); agent.read_info()
    for n in (10, 30, 100, 1000):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(n): agent.update_from_dataset(B)
            t1 = time.perf_counter()
            agent.read_info(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            cur = ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6)
            best = cur if best is None or cur[1] < best[1] else best
        print('%s n=%4d enqueue %.1f us/update, fenced total %.1f us/update' % (prec, n, best[0], best[1]), flush=True)
    agent.close()
