"""Host enqueue time vs device time of the visual update."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import fql_amd
B = 256
cfg = fql_amd.get_config(); cfg.update(encoder='impala_small', alpha=300.0, batch_size=B)
rng = np.random.default_rng(0)
n = 2000
ds = {'observations': rng.integers(0, 256, size=(n, 64, 64, 3), dtype=np.uint8), 'next_observations': rng.integers(0, 256, size=(n, 64, 64, 3), dtype=np.uint8),
      'actions': rng.uniform(-1, 1, size=(n, 5)).astype(np.float32), 'rewards': -np.ones(n, np.float32), 'masks': np.ones(n, np.float32),
      'terminals': np.zeros(n, np.float32)}
torch.zeros(1, device='cuda')
agent = fql_amd.FQLAgent.create(0, np.zeros((1, 64, 64, 9), np.uint8), ds['actions'][:1], cfg)
agent.upload_dataset(ds, frame_stack=3, p_aug=0.5)
for _ in range(5): agent.update_from_dataset(B)
torch.cuda.synchronize()
for label in ('a', 'b'):
    t0 = time.perf_counter()
    for _ in range(20): agent.update_from_dataset(B)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
