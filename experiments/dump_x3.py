import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, numpy as np
import fql_amd
cfg = fql_amd.get_config(); cfg['alpha']=10.0; cfg['precision']='bf16x3'
try:
    a = fql_amd.FQLAgent.create(0, np.zeros((1,29),np.float32), np.zeros((1,8),np.float32), cfg)
    a.update_from_dataset if False else None
    from fql_amd.synthetic import make_synthetic_dataset
    ds = make_synthetic_dataset(10000, 29, 8, seed=0); a.upload_dataset(ds); a.update_from_dataset(256)
except Exception as e:
    print('ERR', str(e)[:300])
