import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd
from fql_amd import _cabi
from tests.util import make_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg, ds, batch, noise = make_problem(29, 8, B, (512,) * 4, seed=3)
a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
lib = _cabi.load()
f = lib.fql_debug_xcd_err; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p]
e = C.c_uint()
for i in range(3):
    t0 = time.perf_counter()
    _, info = a.update(batch, noise=noise)
    v = float(info['critic/critic_loss'])
    dt = time.perf_counter() - t0
    f(a._h, C.byref(e))
    print('update', i, 'took %.3f s' % dt, 'critic_loss', v, 'err', e.value)
