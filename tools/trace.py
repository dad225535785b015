#!/usr/bin/env python3
"""Per-launch timeline of the last full update found in a rocprofv3 kernel-trace CSV (sorted by start)."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(files[0])) if 'fql_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'finalize' in r['Kernel_Name']]
s = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(s[0]['Start_Timestamp'])
agg = {}
end = t0
for r in s:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    end = max(end, en)
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if '-v' in sys.argv:
        print(f"q{r['Queue_Id']:>2s} {name[:24]:24s} grid={int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):5d} start={(st-t0)/1e3:8.1f} end={(en-t0)/1e3:8.1f} dur={(en-st)/1e3:6.1f}")
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += (en - st) / 1e3
print('step total us', (end - t0) / 1e3, 'launches', len(s))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:28s} n={n:3d} total={t:8.1f}us avg={t/n:6.1f}us')
