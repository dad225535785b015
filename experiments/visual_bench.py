"""Timing of the visual update (BASELINE config 5 shape: 64x64x9 uint8, impala_small, B=256, alpha=300)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import fql_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = fql_amd.get_config(); cfg.update(encoder='impala_small', alpha=300.0, batch_size=B)
rng = np.random.default_rng(0)
batch = {'observations': rng.integers(0, 256, size=(B, 64, 64, 9), dtype=np.uint8),
         'next_observations': rng.integers(0, 256, size=(B, 64, 64, 9), dtype=np.uint8),
         'actions': rng.uniform(-1, 1, size=(B, 5)).astype(np.float32),
         'rewards': -np.ones(B, np.float32), 'masks': np.ones(B, np.float32)}
agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
dev = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
for _ in range(3): agent.update(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): _, info = agent.update(dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('visual B=%d: %.3f ms/update, %.1f steps/s' % (B, dt * 1e3, 1 / dt), {k: round(float(v), 4) for k, v in info.items()}, agent.stats())
