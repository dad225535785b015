import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import fql_amd
from fql_amd import _cabi
from oracle import fql_oracle as O, encoder_oracle as E
from tests.test_gpu_visual import make_visual
from tests.util import randomize_params, leaf_dict
cfg, batch, _ = make_visual()
B, ad = 32, 4
agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
params = randomize_params(agent.get_params(), seed=3, scale=0.05)
agent.set_params(params)
nz = O.make_noise(B, ad, 50)
agent.total_loss(batch, None, noise=nz)
lib = _cabi.load()
lib.fql_debug_enc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
def rd(enc, code, shape, dt=np.float32):
    a = np.empty(shape, dt)
    rc = lib.fql_debug_enc(agent._h, enc, code, a.ctypes.data, a.nbytes); assert rc == 0, rc
    return a
for enc, name, n in [(1, 'modules_actor_bc_flow', 32), (2, 'modules_actor_onestep_flow', 64)]:
    imgs = batch['observations'] if n == 32 else np.concatenate([batch['observations'], batch['next_observations']])
    p = O.tree_map(lambda a: a.astype(np.float64), params[name]['encoder'])
    out, cache = E.impala_forward(p, imgs, keep=True, dtype=np.float64)
    res = 32
    for s in range(3):
        sc = cache['stacks'][s]
        C_ = [16, 32, 32][s]
        c0_ref = E.conv3x3(sc['x0'], p[f'stack_blocks_{s}']['Conv_0']['kernel'], p[f'stack_blocks_{s}']['Conv_0']['bias'])
        c0 = rd(enc, 100 * s + 0, (n, res, res, C_))
        pool_ref, arg_ref = E.max_pool(c0_ref)
        pool = rd(enc, 100 * s + 1, (n, res // 2, res // 2, C_))
        arg = rd(enc, 100 * s + 4, (n, res // 2, res // 2, C_), np.uint8)
        inp, c1_ref = sc['blocks'][0]
        c1 = rd(enc, 100 * s + 2, (n, res // 2, res // 2, C_))
        y = rd(enc, 100 * s + 3, (n, res // 2, res // 2, C_))
        y_ref = cache['stacks'][s + 1]['x0'] if s < 2 else cache['final']
        print(name, 'stack', s, 'c0 %.1e pool %.1e argmis %d c1 %.1e y %.1e' % (
            np.abs(c0 - c0_ref).max(), np.abs(pool - pool_ref).max(), int((arg != arg_ref).sum()), np.abs(c1 - c1_ref).max(), np.abs(y - y_ref).max()))
        res //= 2
    Eg = rd(enc, 902, (n, 512))
    print(name, 'E %.1e' % np.abs(Eg - out).max())
