"""What precision = 'bf16x3' (fql_config.precision = 2) costs in accuracy: one update at BASELINE configs[1] and at a small ragged
shape against the fp64 oracle, beside the fp32 engine on the same inputs.  Prints per-info deltas and the worst per-leaf gradient
error (max|g - g_ref| / max|g_ref|); the bounds in tests/test_gpu_precision.py come from this output."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd  # noqa: E402
from oracle import fql_oracle as O  # noqa: E402
from tests.util import leaf_dict, make_problem, randomize_params  # noqa: E402


def probe(od, ad, B, hidden, alpha, steps=1):
    out = {}
    for prec in ('fp32', 'bf16x3'):
        cfg, ds, batch, noise = make_problem(od, ad, B, hidden, seed=41, alpha=alpha)
        cfg['precision'] = prec
        agent = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
        params = randomize_params(agent.get_params(), seed=9, scale=0.05)
        agent.set_params(params)
        ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
        _, _, g_ref = ref.grads(batch, noise)
        _, info = agent.update(batch, noise=noise)
        _, info_r = ref.update(batch, noise)
        info = dict(info)
        mu = leaf_dict(agent.get_opt_state()['mu'])
        worst = (0.0, None)
        for p, g in leaf_dict(g_ref).items():
            sc = np.abs(g).max()
            err = np.abs(mu[p] / 0.1 - g).max() / max(sc, 1e-30)
            worst = max(worst, (float(err), p))
        d = {k: (float(info[k]) - float(info_r[k])) for k in O.INFO_KEYS}
        rel = {k: abs(d[k]) / max(1.0, abs(float(info_r[k]))) for k in d}
        out[prec] = (worst, d, rel)
        print(f'--- {prec} od={od} ad={ad} B={B} hidden={hidden} alpha={alpha}')
        print('   worst per-leaf gradient error / max|g|: %.3e (%s)' % worst)
        for k in O.INFO_KEYS:
            print('   %-22s ref % .7e  delta % .3e  rel-to-max(1,|ref|) %.3e' % (k, float(info_r[k]), d[k], rel[k]))
    return out


if __name__ == '__main__':
    probe(29, 8, 256, (512, 512, 512, 512), 10.0)
    probe(40, 4, 1024, (512, 512, 512, 512), 300.0)
    probe(29, 8, 64, (64, 64, 64, 64), 10.0)
    probe(17, 6, 32, (80, 48, 64, 32), 10.0)
