"""The opt-in XCD-resident Euler chain (FQL_XCHAIN=1, fql_amd/csrc/fql_xchain.h): one persistent launch instead of 30 chain launches.

It is off by default (measured slower end to end at B = 256: profiles/r03_xcd_resident.txt), so it gets its own parity test: the update with
the persistent chain against the fp64 oracle (infos, per-leaf gradients through Adam's first moment) and against the default program on
identical inputs (the Euler target feeds the distillation loss, so the one-step actor's gradients see every step of the chain)."""
import os

import numpy as np
import pytest

from oracle import fql_oracle as O
from tests.util import assert_info_close, leaf_dict, make_problem, randomize_params

pytestmark = pytest.mark.gpu


def _run(cfg, batch, noise, params, xchain):
    import fql_amd
    if xchain:
        os.environ['FQL_XCHAIN'] = '1'
    else:
        os.environ.pop('FQL_XCHAIN', None)
    try:
        a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    finally:
        os.environ.pop('FQL_XCHAIN', None)
    a.set_params(params)
    launches = a.stats()['launches_per_update']
    _, info = a.update(batch, noise=noise)
    info = {k: float(info[k]) for k in O.INFO_KEYS}
    return a, info, launches


@pytest.mark.parametrize('B,H,fs', [(128, 256, 3), (256, 512, 10)])
def test_xchain_update_matches_oracle_and_default_program(B, H, fs):
    od, ad = 29, 8
    cfg, ds, batch, noise = make_problem(od, ad, B, (H,) * 4, seed=3, flow_steps=fs)
    import fql_amd
    base = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
    params = randomize_params(base.get_params(), seed=5)
    a0, i0, l0 = _run(cfg, batch, noise, params, False)
    a1, i1, l1 = _run(cfg, batch, noise, params, True)
    assert l1 < l0, (l0, l1)          # the persistent launch really replaced the chain's launches
    ref = O.OracleFQL(params, dict(cfg), od, ad, np.float64)
    _, _, g_ref = ref.grads(batch, noise)
    _, info_ref = ref.update(batch, noise)
    assert_info_close(i1, info_ref, rtol=5e-5, atol=5e-6)
    mu0, mu1 = leaf_dict(a0.get_opt_state()['mu']), leaf_dict(a1.get_opt_state()['mu'])
    for p, g in leaf_dict(g_ref).items():
        tol = 2e-5 * np.abs(g).max() + 1e-9
        np.testing.assert_allclose(mu1[p] / 0.1, g, rtol=0, atol=tol, err_msg=f'grad vs oracle {p}')
        np.testing.assert_allclose(mu1[p], mu0[p], rtol=0, atol=0.1 * tol, err_msg=f'grad vs default program {p}')
    # the sticky error word of the persistent launch stays clear (a placement it cannot run on, or a wait that timed out, would set it)
    _, info2 = a1.update(batch, noise=noise)
    assert np.isfinite(float(info2['actor/distill_loss']))
