"""Phase timeline of the XCD-resident launch from a -DFQL_XSTAMPS build (FQL_AMD_LIB=experiments/libfql_xst.so).
Per phase: mean over the 256 workgroups of wait (previous arrive -> wait over), ops, drain; and the phase's span over all workgroups."""
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
for _ in range(5):
    a.update_from_dataset(B)
a.read_info()
lib = _cabi.load()
f = lib.fql_debug_xcd_stamps
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
buf = np.zeros(256 * 128 * 4 + 256 * 16, dtype=np.uint64)
nph = C.c_int()
rc = f(a._h, buf.ctypes.data, buf.size, C.byref(nph))
assert rc == 0, rc
P = nph.value
st = buf[:256 * P * 4].reshape(256, P, 4).astype(np.int64)
ids = st[:, 0, 3].copy()
t0 = st[:, :, 0].min()
print('phases', P, ' total span %.1f us' % ((st[:, :, 2].max() - t0) / 100.0))
for p in range(P):
    wait = (st[:, p, 0] - (st[:, p - 1, 2] if p else st[:, p, 0])).mean() / 100.0
    ops = (st[:, p, 1] - st[:, p, 0]).mean() / 100.0
    opsmax = (st[:, p, 1] - st[:, p, 0]).max() / 100.0
    drain = (st[:, p, 2] - st[:, p, 1]).mean() / 100.0
    span = (st[:, p, 2].max() - st[:, p, 0].min()) / 100.0
    print(f'phase {p:2d}: start {((st[:, p, 0].min() - t0) / 100.0):7.1f}  wait {wait:6.2f}  ops mean {ops:6.2f} max {opsmax:6.2f}  drain {drain:5.2f}  span {span:6.2f}')

if os.environ.get('FQL_XSTAMP_PHASE'):
    s2 = buf[256 * P * 4:256 * P * 4 + 256 * 16].reshape(256, 16).astype(np.int64)
    names = ['op loaded', 'col tile start', 'A + W loads done', 'prologue done', 'MFMA done', 'reduce sync', 'epilogue done', 'phase ops done']
    base = s2[:, 8]
    print('inside phase', os.environ['FQL_XSTAMP_PHASE'], '(last op of the phase; mean us since phase start over workgroups):')
    for k in range(8):
        print(f'   {names[k]:20s} {((s2[:, k] - base).mean() / 100.0):7.2f}')
