"""Is the run-to-run spread of the bf16x3 rate (2830 - 3270 updates/s over processes on one box, fp32: +- 0.3 %) a property of the process
or does the rate change inside one?  Several agents in one process, several 500-update segments each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset
B = 256
ds = make_synthetic_dataset(100000, 29, 8, seed=0)
torch.zeros(1, device='cuda')
for prec in ('bf16x3', 'bf16x3', 'fp32', 'bf16x3'):
    cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=B, precision=prec)
    agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
    agent.upload_dataset(ds)
    for _ in range(300): agent.update_from_dataset(B)
    agent.read_info(); torch.cuda.synchronize()
    out = []
    for seg in range(8):
        t0 = time.perf_counter()
        for _ in range(500): agent.update_from_dataset(B)
        agent.read_info(); torch.cuda.synchronize()
        out.append(500 / (time.perf_counter() - t0))
    print(prec, ' '.join('%.0f' % v for v in out), flush=True)
    agent.close()
