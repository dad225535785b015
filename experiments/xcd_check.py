"""A/B of the XCD-resident update (fql_xcd.h) against the launch-per-level program on identical parameters, batch and noise:
13 infos, Adam first moments (= 0.1 x gradient after the first step from zero moments) and post-step parameters, per leaf.
usage: python experiments/xcd_check.py [B] [H] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd  # noqa: E402
from oracle import fql_oracle as O  # noqa: E402
from tests.util import make_problem, randomize_params, leaf_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
od, ad = 29, 8
cfg, ds, batch, noise = make_problem(od, ad, B, (H,) * 4, seed=3)


def run(no_xcd):
    if no_xcd:
        os.environ.pop('FQL_XCHAIN', None)
    else:
        os.environ['FQL_XCHAIN'] = '1'
    a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    a.set_params(randomize_params(a.get_params(), 5))
    infos = []
    for s in range(steps):
        _, info = a.update(batch, noise=noise)
        infos.append({k: float(info[k]) for k in O.INFO_KEYS})
    st = a.get_opt_state()
    return a, infos, leaf_dict(a.get_params()), leaf_dict(st['mu'])


a0, i0, p0, m0 = run(True)
print('launches (old):', a0.stats())
a1, i1, p1, m1 = run(False)
print('launches (xcd):', a1.stats())
for s in range(steps):
    for k in O.INFO_KEYS:
        d = abs(i0[s][k] - i1[s][k])
        flag = '' if d <= 2e-6 + 2e-5 * abs(i0[s][k]) else '   <-- DIFF'
        print(f'step {s} {k:24s} old {i0[s][k]: .7e} xcd {i1[s][k]: .7e} |d| {d:.2e}{flag}')
worst = 0.0
for name in sorted(m0):
    g0, g1 = m0[name], m1[name]
    sc = max(np.abs(g0).max(), 1e-30)
    e = np.abs(g0 - g1).max() / sc
    worst = max(worst, e)
    pe = np.abs(p0[name] - p1[name]).max()
    flag = '' if e <= 2e-5 else '   <-- DIFF'
    print(f'{name:60s} mu rel err {e:.2e}  param abs err {pe:.2e}{flag}')
print('worst relative first-moment error:', worst)
# timing
for name, a in (('old', a0), ('xcd', a1)):
    a.upload_dataset(ds)
    for _ in range(20):
        a.update_from_dataset(B)
    a.read_info()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        a.update_from_dataset(B)
    a.read_info()
    dt = time.perf_counter() - t0
    print(f'{name}: {n / dt:.0f} updates/s ({dt / n * 1e6:.1f} us per update)')
