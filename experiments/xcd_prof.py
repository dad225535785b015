"""Per-launch device time of the update program in use (fql_profile_update) + the XCD program's phase dump (FQL_DUMP=1).
usage: python experiments/xcd_prof.py [B] [H]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd  # noqa: E402
from fql_amd import _cabi  # noqa: E402
from tests.util import make_problem  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg, ds, batch, noise = make_problem(29, 8, B, (H,) * 4, seed=3)
a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
a.upload_dataset(ds)
lib = _cabi.load()
f = lib.fql_profile_update
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
cap = 256
typ, lane, grid = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
us, macs, nul = (C.c_float * cap)(), (C.c_double * cap)(), C.c_float()
for r in range(4):
    n = f(a._h, B, cap, typ, lane, grid, us, macs, C.byref(nul))
print('launches', n)
for i in range(n):
    print(f'  #{i} type {typ[i]:2d} grid {grid[i]:5d}  {us[i]:8.1f} us   {2 * macs[i] / max(us[i], 1e-3) / 1e6:7.2f} TFLOP/s')
