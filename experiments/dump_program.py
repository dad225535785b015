"""FQL_DUMP of the levelled programs: python experiments/dump_program.py [visual]"""
import os
import sys
os.environ['FQL_DUMP'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import fql_amd  # noqa: E402
from tests.util import make_problem  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] == 'visual':
    from fql_amd.synthetic import make_synthetic_frames
    cfg = fql_amd.get_config()
    cfg.update(alpha=300.0, batch_size=256, encoder='impala_small')
    ds = make_synthetic_frames(600, 5, seed=0)
    a = fql_amd.FQLAgent.create(0, np.zeros((1, 64, 64, 9), np.uint8), ds['actions'][:1], cfg)
else:
    cfg, ds, batch, noise = make_problem(29, 8, 256, (512,) * 4, seed=3)
    a = fql_amd.FQLAgent.create(0, batch['observations'][:1], batch['actions'][:1], cfg)
