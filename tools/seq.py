"""Per-launch durations (in issue order) of the encoder kernels of the last traced update; arg: rocprofv3 output dir."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fin = [i for i, r in enumerate(rows) if 'fql_finalize' in r['Kernel_Name']]
seq = rows[fin[-2] + 1:fin[-1] + 1]
out, tot = [], {}
for r in seq:
    n = r['Kernel_Name']
    short = ('conv' if 'conv3x3' in n else 'wgrad' if 'conv_wgrad_kernel' in n else 'wred' if 'reduce' in n else
             'pool' if 'maxpool_k' in n else 'poolb' if 'maxpool_bwd' in n else 'other')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[short] = tot.get(short, 0) + d
    if short != 'other': out.append('%s:%.0f' % (short, d))
print(len(seq), ' '.join(out))
print({k: round(v) for k, v in tot.items()}, 'sum', round(sum(tot.values())))
