"""Phase timeline of the XCD-resident launch from a -DFQL_XSTAMPS build (FQL_AMD_LIB=experiments/libfql_xst.so), per team.
Per phase: mean over the 256 workgroups of wait (previous arrive -> wait over), ops, drain; and the phase's start.
FQL_XSTAMP_PHASE=<p> FQL_XSTAMP_TEAM=<t>: stamps inside the last op of that phase."""
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
buf = np.zeros(1024 * 128 * 4 + 256 * 25, dtype=np.uint64)
nph = C.c_int()
rc = f(a._h, buf.ctypes.data, buf.size, C.byref(nph))
assert rc == 0, rc
P = nph.value
st = buf[:1024 * P * 4].reshape(256, 4, P, 4).astype(np.int64)
t0 = st[:, :, :, 0][st[:, :, :, 0] > 0].min()
t1 = st[0, 2]
v = np.nonzero(t1[:, 2] > 0)[0]
if len(v) > 1:
    print('shader clock over team 2 of workgroup 0: %.0f MHz' % ((t1[v[-1], 3] - t1[v[0], 3]) / ((t1[v[-1], 2] - t1[v[0], 2]) / 100.0)))
for team in range(4):
    s = st[:, team]
    if not (s[0, :, 2] > 0).any():
        continue
    np_t = int(np.nonzero(s[0, :, 2] > 0)[0].max()) + 1
    print(f'team {team}: {np_t} phases, ends at {((s[:, :np_t, 2].max() - t0) / 100.0):.1f} us')
    for p in range(np_t):
        if (s[:, p, 0] <= 0).any():
            continue
        wait = (s[:, p, 0] - (s[:, p - 1, 2] if p and (s[:, p - 1, 2] > 0).all() else s[:, p, 0])).mean() / 100.0
        ops = (s[:, p, 1] - s[:, p, 0]).mean() / 100.0
        opsmax = (s[:, p, 1] - s[:, p, 0]).max() / 100.0
        drain = (s[:, p, 2] - s[:, p, 1]).mean() / 100.0
        print(f'  phase {p:2d}: start {((s[:, p, 0].min() - t0) / 100.0):7.1f}  wait {wait:6.2f}  ops mean {ops:6.2f} max {opsmax:6.2f}  drain {drain:5.2f}')
if os.environ.get('FQL_XSTAMP_PHASE'):
    s2 = buf[1024 * P * 4:1024 * P * 4 + 256 * 16].reshape(256, 16).astype(np.int64)
    names = ['op loaded', 'col tile start', 'A + W loads done', 'prologue done', 'MFMA done', 'reduce sync', 'epilogue done', 'phase ops done']
    base = s2[:, 8]
    print('inside phase', os.environ['FQL_XSTAMP_PHASE'], '(last op of the phase; mean us since phase start over workgroups):')
    for k in range(8):
        print(f'   {names[k]:20s} {((s2[:, k] - base).mean() / 100.0):7.2f}')
    for k, nm in ((9, 'fold start'), (10, 'fold loads issued'), (11, 'fold reduced'), (12, 'fold done')):
        if (s2[:, k] > 0).all():
            print(f'   {nm:20s} {((s2[:, k] - base).mean() / 100.0):7.2f}')

tw = buf[1024 * P * 4 + 256 * 17:1024 * P * 4 + 256 * 25].reshape(256, 8).astype(np.int64)
n = np.maximum(tw[:, 4], 1)
print('team 0 waits, mean us per wait over workgroups: poll %.2f  barrier after poll %.2f | per arrive: drain %.2f  barrier %.2f  (%d waits)' % (
    (tw[:, 0] / n).mean() / 100, (tw[:, 1] / n).mean() / 100, (tw[:, 2] / n).mean() / 100, (tw[:, 3] / n).mean() / 100, int(n.mean())))
