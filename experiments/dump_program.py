import os, sys
os.environ['FQL_DUMP']='1'
sys.path.insert(0, os.getcwd())
import numpy as np, fql_amd
from tests.util import make_problem
cfg, ds, batch, noise = make_problem(29, 8, 256, (512,)*4, seed=3)
a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
a.update(batch, noise=noise)
