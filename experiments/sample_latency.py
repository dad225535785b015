"""Latency of sample_actions / compute_flow_actions for online acting (main.py:225: one observation per env step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fql_amd
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=256)
a = fql_amd.FQLAgent.create(0, np.zeros((1, 29), np.float32), np.zeros((1, 8), np.float32), cfg)
for n in (1, 16, 256):
    obs = np.random.randn(n, 29).astype(np.float32)
    z = np.random.randn(n, 8).astype(np.float32)
    for f, name in ((lambda: a.sample_actions(obs, seed=1), 'sample_actions(rng)'), (lambda: a.sample_actions(obs, noises=z), 'sample_actions(noise)'),
                    (lambda: a.compute_flow_actions(obs, z), 'compute_flow_actions')):
        for _ in range(20): f()
        t = time.perf_counter()
        for _ in range(200): f()
        print(f'n={n:4d} {name:24s} {(time.perf_counter() - t) / 200 * 1e6:8.1f} us per call (host numpy in/out, synchronous)')
