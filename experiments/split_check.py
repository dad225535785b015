import sys; sys.path.insert(0, '.')
import numpy as np, torch
import fql_amd
from oracle import fql_oracle as O
torch.zeros(1, device='cuda')
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=256)
ds = O.make_synthetic_dataset(10000, 29, 8, seed=0)
a = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
a.upload_dataset(ds)
print('buckets', a.grad_buckets())
print('default stream handle', torch.cuda.current_stream().cuda_stream)
s1 = torch.cuda.Stream()
try:
    a.update_begin_split(torch.cuda.current_stream().cuda_stream, s1.cuda_stream, batch_size=256)
    a.update_end(stream=torch.cuda.current_stream().cuda_stream)
    print('split with default stream ok')
except Exception as e:
    print('default stream failed:', e)
s0 = torch.cuda.Stream()
a.update_begin_split(s0.cuda_stream, s1.cuda_stream, batch_size=256); s0.wait_stream(s1); a.update_end(stream=s0.cuda_stream)
torch.cuda.synchronize(); print('explicit streams ok', a.read_info()['critic/critic_loss'])
