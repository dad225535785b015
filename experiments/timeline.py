#!/usr/bin/env python3
"""Concurrent (un-profiled) timeline of ONE update from the -DFQL_TIMELINE build: per launch, lane, op type, workgroups, first / last
workgroup entry and last exit in us of device wall clock.  Usage:
    FQL_AMD_LIB=experiments/libfql_tl.so python experiments/timeline.py [n_updates_to_average_over]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
if os.environ.get("FQL_TL_TORCH"):
    import torch  # noqa: F401  (use the HIP runtime bundled with torch, as bench.py does)

import fql_amd  # noqa: E402
from fql_amd import _cabi  # noqa: E402
from fql_amd.synthetic import make_synthetic_dataset  # noqa: E402

TYPES = ['gemm16', 'side', 'wgrad', 'lnbwd', 'prep', 'postos', 'euler_fin', 'pec', 'loss_critic', 'loss_q', 'loss_bc', 'loss_actor', 'conv_wprep',
         'conv', 'conv_u8', 'pool', 'pool_bwd', 'conv_wgrad', 'conv_wred', 'enc_dz', 'chain', 'wfrag', 'xchain', 'head_dgrad', 'dgrad0', 'adam', 'finalize']
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
od, ad, B = 29, 8, 256
cfg = fql_amd.get_config()
cfg.update(alpha=10.0, batch_size=B, precision=os.environ.get('FQL_TL_PRECISION', 'fp32'))
ds = make_synthetic_dataset(100_000, od, ad, seed=0)
agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
agent.upload_dataset(ds)
lib = _cabi.load()
f = lib.fql_debug_timeline
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
cap = 256
lane, typ, grid = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
t0, t1, t2 = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
for _ in range(200):
    agent.update_from_dataset(B)
acc = None
for r in range(reps):
    for _ in range(20):
        agent.update_from_dataset(B)
    assert f(agent._h, 1, cap, lane, typ, grid, t0, t1, t2) == 0      # (synchronises: the device is idle here)
    # FQL_TL_STEADY=K: K updates back to back, the stamps that remain are the LAST one's (every launch overwrites its slots) - an update in
    # steady state, its launches enqueued while the previous update was still running.  Default: ONE update launched on an idle device.
    for _ in range(int(os.environ.get('FQL_TL_STEADY', '1'))):
        agent.update_from_dataset(B)
    where = agent.synchronize()
    n = f(agent._h, 0, cap, lane, typ, grid, t0, t1, t2)
    assert n > 0, n
    cur = np.array([[t0[i], t1[i], t2[i]] for i in range(n)])
    acc = cur if acc is None else acc + cur
acc /= reps
order = np.argsort(acc[:, 0])
print(f'# {n} launches ({where} dispatch), mean over {reps} updates; us since the first workgroup entry of the update')
print('# lane  op            wgs   first_entry  last_entry  last_exit   span')
for i in order:
    if grid[i] == 0:
        continue
    print(f'  {lane[i]}    {TYPES[typ[i]]:12s} {grid[i]:5d}  {acc[i,0]:10.1f} {acc[i,1]:10.1f} {acc[i,2]:10.1f} {acc[i,2]-acc[i,0]:7.1f}')
print(f'# update ends at {acc[:, 2].max():.1f} us')
