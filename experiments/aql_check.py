"""The update on the engine's own AQL queues (fql_amd/csrc/fql_aql.h): one step against the fp64 oracle, then the rate of a short and
a long window, and the same with FQL_AQL=0 in a child process (the captured graph).
usage: python experiments/aql_check.py [B] [H] [only]     (only: this process alone, whatever FQL_AQL says)"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd  # noqa: E402
from oracle import fql_oracle as O  # noqa: E402
from tests.util import make_problem, randomize_params, assert_step_matches  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
child = len(sys.argv) > 3 and sys.argv[3] == 'child'
prec = os.environ.get('FQL_CHECK_PRECISION', 'fp32')
od, ad = 29, 8
cfg, ds, batch, noise = make_problem(od, ad, B, (H,) * 4, seed=3)
cfg['precision'] = prec
tag = ('graph' if os.environ.get('FQL_AQL') == '0' else 'aql') + ' ' + prec

a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
a.set_params(randomize_params(a.get_params(), 5))
a.upload_dataset(ds)
ref = O.OracleFQL(a.get_params(), dict(cfg), od, ad, np.float64)
idxs = np.arange(B)
b0 = O.sample_batch(ds, idxs)


def step(agent, bt, nz):
    agent.update_from_dataset(B, idxs=idxs, noise=nz)
    print(f'[{tag}] the checked update ran on:', agent.synchronize())
    return None


if prec == 'fp32' and not os.environ.get('FQL_CHECK_SKIP'):
    worst = assert_step_matches(a, ref, cfg, b0, noise, step=step)
    print(f'[{tag}] one update against the fp64 oracle: worst relative gradient error {worst[0]:.2e} ({worst[1]})')
for n, warm in ((20, 3), (20, 3), (300, 20), (1000, 20)):
    for _ in range(warm):
        a.update_from_dataset(B)
    a.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        a.update_from_dataset(B)
    t1 = time.perf_counter()
    where = a.synchronize()
    t2 = time.perf_counter()
    print(f'[{tag}] {where}: n={n:5d} enqueue {(t1 - t0) / n * 1e6:7.1f} us/update, fenced {(t2 - t0) / n * 1e6:7.1f} us/update = {n / (t2 - t0):7.1f} updates/s')
info = a.read_info()
print(f'[{tag}] last info:', {k: round(v, 5) for k, v in list(info.items())[:4]})
if len(sys.argv) <= 3:
    env = dict(os.environ, FQL_AQL='0')
    sys.exit(subprocess.run([sys.executable, __file__, str(B), str(H), 'child'], env=env).returncode)
