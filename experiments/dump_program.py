import os, numpy as np
os.environ['FQL_DUMP']='1'
import fql_amd
cfg = fql_amd.get_config(); cfg['alpha']=10.0
a = fql_amd.FQLAgent.create(0, np.zeros((1,29),np.float32), np.zeros((1,8),np.float32), cfg)
