"""FQL_TRACE=1 python experiments/trace_waits.py 2> waits.txt : the launch list of the fused update program with every cross-lane wait."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset
ds = make_synthetic_dataset(10000, 29, 8, seed=0)
cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=256, precision=os.environ.get('PREC', 'fp32'))
agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
agent.upload_dataset(ds)
agent.update_from_dataset(256)
agent.read_info()
agent.close()
