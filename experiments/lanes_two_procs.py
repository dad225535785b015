"""Experiment: hardware interference between the two lanes, measured without any graph-branch machinery:
two engines in one process, one runs only lane 0 of its update, the other only lane 1, on separate streams."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fql_amd
from oracle import fql_oracle as O
od, ad, B = 29, 8, 256
ds = O.make_synthetic_dataset(100000, od, ad, seed=0)
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=B)
def mk():
    a = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg); a.upload_dataset(ds); return a
a0 = mk()
N = 500
def run(a, n):
    for _ in range(n): a.update_from_dataset(B)
run(a0, 50); torch.cuda.synchronize()
t = time.perf_counter(); run(a0, N); torch.cuda.synchronize(); print('single engine, lane', os.environ.get('FQL_ONLY_LANE'), ':', (time.perf_counter() - t) / N * 1e6, 'us/step')
