#!/usr/bin/env python3
"""Print the per-launch timeline of the last full update found in a rocprofv3 kernel-trace CSV."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(files[0])) if r['Kernel_Name'].startswith('fql_')]
idx = [i for i, r in enumerate(rows) if 'prep' in r['Kernel_Name']]
s = rows[idx[-2]:idx[-1]]
t0 = int(s[0]['Start_Timestamp'])
prev = t0
agg = {}
for r in s:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0]
    if '-v' in sys.argv:
        print(f"{name[:26]:26s} grid={int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):5d} vgpr={r['VGPR_Count']:>3s} start={(st-t0)/1e3:8.1f} dur={(en-st)/1e3:7.1f} gap={(st-prev)/1e3:6.1f}")
    prev = en
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += (en - st) / 1e3
print('step total us', (prev - t0) / 1e3, 'launches', len(s))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:28s} n={n:3d} total={t:8.1f}us avg={t/n:6.1f}us')
