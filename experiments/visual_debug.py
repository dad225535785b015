import sys, numpy as np
sys.path.insert(0, '.')
import fql_amd
from oracle import fql_oracle as O
from tests.test_gpu_visual import make_visual
from tests.util import randomize_params, leaf_dict
cfg, batch, _ = make_visual()
import os
if os.environ.get('SAME'): batch['next_observations'] = batch['observations'].copy()
B, ad = 32, 4
agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
params = randomize_params(agent.get_params(), seed=3, scale=0.05)
agent.set_params(params)
ref = O.OracleFQL(params, dict(cfg), (32, 32, 3), ad, np.float64)
nz = O.make_noise(B, ad, 50)
_, _, g_ref = ref.grads(batch, nz)
agent.update(batch, noise=nz)
mu = leaf_dict(agent.get_opt_state()['mu'])
for p, g in leaf_dict(g_ref).items():
    if p.startswith('modules_target'): continue
    s = np.abs(g).max()
    err = np.abs(mu[p] / 0.1 - g).max() / (s + 1e-30)
    print('%.2e %s' % (err, p))
