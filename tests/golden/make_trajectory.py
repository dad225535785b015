#!/usr/bin/env python3
"""Generates tests/golden/traj_configs1.npz: a 100-update trajectory at BASELINE.json configs[1] / configs[0]
(antmaze-large shape: obs 29, act 8, batch 256, hidden 512x4, flow_steps 10, alpha 10) from the fp32 torch-autograd CPU
restatement of the reference update (oracle/fql_oracle_torch.py).  Run from the repo root:

    python tests/golden/make_trajectory.py

This is the loss-trajectory fixture SURVEY.md 8(d) / BASELINE.md section 4 ask for ("the 13 metrics as the loss-delta fixture"),
at 100 steps so it regenerates in ~30 s of CPU.  It pins the ORACLE, not JAX (parity unpinned at the JAX boundary).  Only seeds
travel: parameters come from oracle.init_params(SEED_PARAMS), the dataset from make_synthetic_dataset(ROWS, seed=SEED_DATA), the
index stream from default_rng(SEED_IDX).integers (utils/datasets.py:66), the five noise tensors of update s from
make_noise(B, act, SEED_NOISE + s) - every consumer regenerates the identical inputs from numpy's stable streams.
Stored: infos [STEPS, 13] (fp64 copies of the fp32 scalars), per-module parameter checksums after the last update.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fql_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OD, AD, B, STEPS, ROWS = 29, 8, 256, 100, 8192
SEED_PARAMS, SEED_DATA, SEED_IDX, SEED_NOISE = 123, 5, 7, 1000


def problem():
    cfg = O.get_config()
    cfg.update(alpha=10.0, batch_size=B)
    params = O.init_params(SEED_PARAMS, OD, AD, cfg)
    ds = O.make_synthetic_dataset(ROWS, OD, AD, seed=SEED_DATA)
    return cfg, params, ds


def inputs(ds):
    rng = np.random.default_rng(SEED_IDX)
    for s in range(STEPS):
        idx = rng.integers(0, ROWS, size=B)
        yield s, O.sample_batch(ds, idx), O.make_noise(B, AD, SEED_NOISE + s)


def main():
    import torch
    from oracle.fql_oracle_torch import TorchFQL
    torch.manual_seed(0)
    cfg, params, ds = problem()
    ref = TorchFQL(params, dict(cfg), torch.float32)
    infos = np.zeros((STEPS, len(O.INFO_KEYS)))
    for s, batch, noise in inputs(ds):
        _, info = ref.update(batch, noise)
        infos[s] = [info[k] for k in O.INFO_KEYS]
        if s % 10 == 0:
            print(s, {k: round(float(v), 5) for k, v in list(info.items())[:5]}, flush=True)
    final = ref.get_params() if hasattr(ref, 'get_params') else ref.params
    sums = {}
    for path, leaf in O.tree_leaves_with_path(final):
        a = np.asarray(leaf.detach().numpy() if hasattr(leaf, 'detach') else leaf, dtype=np.float64)
        mod = path.split('/')[0]
        s0, s1 = sums.get(mod, (0.0, 0.0))
        sums[mod] = (s0 + a.sum(), s1 + (a * a).sum())
    meta = dict(obs_dim=OD, act_dim=AD, B=B, steps=STEPS, rows=ROWS, seeds=dict(params=SEED_PARAMS, data=SEED_DATA, idx=SEED_IDX, noise=SEED_NOISE),
                cfg=dict(alpha=10.0), keys=list(O.INFO_KEYS), torch=torch.__version__, numpy=np.__version__,
                generator='tests/golden/make_trajectory.py (oracle/fql_oracle_torch.py, fp32)')
    np.savez_compressed(os.path.join(HERE, 'traj_configs1.npz'), meta=json.dumps(meta), infos=infos,
                        modules=np.array(sorted(sums)), checksums=np.array([sums[m] for m in sorted(sums)]))
    print('wrote traj_configs1.npz', infos[-1])


if __name__ == '__main__':
    main()
