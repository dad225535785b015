"""A/B of two engine settings in ONE process: two agents, each created under its own environment (the engine reads its knobs when it builds
its programs), timed in alternating fenced windows - process-to-process variance (+-2 %) drops out.
usage: python experiments/ab_inproc.py "KEY=VAL ..." "KEY=VAL ..." [rounds] [updates per window] [precision] [batch]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd  # noqa: E402
from tests.util import make_problem  # noqa: E402

envs = [dict(kv.split('=', 1) for kv in a.split()) if a.strip() and a.strip() != '-' else {} for a in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
n = int(sys.argv[4]) if len(sys.argv) > 4 else 300
prec = sys.argv[5] if len(sys.argv) > 5 else 'fp32'
B = int(sys.argv[6]) if len(sys.argv) > 6 else 256
H = 512
cfg, ds, batch, noise = make_problem(29, 8, B, (H,) * 4, seed=3)
cfg['precision'] = prec
agents = []
from fql_amd import _cabi  # noqa: E402
for e in envs:
    lib = e.pop('LIB', None)          # LIB=<path>: this agent runs another build of the library (both live in the process)
    if lib:
        _cabi.LIB_PATH = os.path.abspath(lib)
        _cabi._lib = None
    e_show = dict(e, **({'LIB': lib} if lib else {}))
    for k, v in e.items():
        os.environ[k] = v
    a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    a.upload_dataset(ds)
    for _ in range(30):
        a.update_from_dataset(B)
    a.synchronize()
    agents.append(a)
    for k in e:
        del os.environ[k]
    e.update(e_show)
res = [[], []]
for r in range(rounds):
    for i, a in enumerate(agents):
        for _ in range(10):
            a.update_from_dataset(B)
        a.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            a.update_from_dataset(B)
        a.synchronize()
        res[i].append((time.perf_counter() - t0) / n * 1e6)
for i, e in enumerate(envs):
    print(f'{"AB"[i]} {e}: median {statistics.median(res[i]):.1f} us/update  (min {min(res[i]):.1f}, max {max(res[i]):.1f})  = {1e6 / statistics.median(res[i]):.0f} updates/s')
print(f'B / A time: {statistics.median(res[1]) / statistics.median(res[0]):.4f}')
