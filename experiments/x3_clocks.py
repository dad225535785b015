"""Do the two steady states of the bf16x3 rate (experiments/x3_modes.py) go with the GPU's clock / power state?  A thread samples the hwmon clock and
power files of every amdgpu device (whichever exist; read-only sysfs) while an agent runs 500-update segments; per segment: rate, mean and min clock, mean power."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fql_amd
from fql_amd.synthetic import make_synthetic_dataset

files = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input')) + sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average')) \
    + sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'))
print('sampling', files, flush=True)
samples, stop = [], False
def sampler():
    while not stop:
        row = [time.perf_counter()]
        for f in files:
            try: row.append(float(open(f).read().strip()))
            except Exception: row.append(float('nan'))
        samples.append(row)
        time.sleep(0.005)
th = threading.Thread(target=sampler, daemon=True); th.start()
B = 256
ds = make_synthetic_dataset(100000, 29, 8, seed=0)
torch.zeros(1, device='cuda')
for prec in ('bf16x3', 'fp32', 'bf16x3'):
    cfg = fql_amd.get_config(); cfg.update(alpha=10.0, batch_size=B, precision=prec)
    agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
    agent.upload_dataset(ds)
    for _ in range(300): agent.update_from_dataset(B)
    agent.read_info(); torch.cuda.synchronize()
    for seg in range(10):
        t0 = time.perf_counter()
        for _ in range(500): agent.update_from_dataset(B)
        agent.read_info(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        rows = [r for r in samples if t0 <= r[0] <= t1]
        cols = list(zip(*rows)) if rows else []
        desc = ' '.join('%s mean %.4g min %.4g max %.4g' % (os.path.basename(f), sum(c) / len(c), min(c), max(c)) for f, c in zip(files, cols[1:])) if rows else 'no samples'
        print('%s seg %d: %.0f updates/s | %d samples | %s' % (prec, seg, 500 / (t1 - t0), len(rows), desc), flush=True)
    agent.close()
stop = True
