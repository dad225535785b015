"""Per-leaf error of the engine's gradient against tests/golden/visual_full.npz's strided sample, in units of the leaf's largest
gradient element.  usage: python experiments/golden_err.py [fp32|bf16x3]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd  # noqa: E402
from oracle import fql_oracle as O  # noqa: E402
from tests.test_golden_oracle import VISUAL_FULL, load_visual_case  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
c = load_visual_case(VISUAL_FULL)
m, z = c['meta'], c['z']
cfg = dict(c['cfg']); cfg['precision'] = prec
a = fql_amd.FQLAgent.create(0, c['batch']['observations'][:1], c['batch']['actions'][:1], cfg)
a.set_params(c['params'])
a.update(c['batch'], noise=c['noise'])
mu = dict(O.tree_leaves_with_path(a.get_opt_state()['mu']))
off = 0
for (p, v), gmax in zip(mu.items(), z['grad_max']):
    g = (v.astype(np.float64) / 0.1).reshape(-1)[::max(1, v.size // 64)][:64]
    ref = z['grad_sample'][off:off + len(g)]; off += len(g)
    if gmax > 0:
        e = np.abs(g - ref) / gmax
        print(f'{p:64s} max {e.max():.2e} median {np.median(e):.2e}')
