#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter CSV per kernel: mean counter value per dispatch and per update."""
import csv
import glob
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
files = glob.glob(path + '/**/*counter_collection.csv', recursive=True)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if not k.startswith('fql_') and not k.startswith('void fql_'):
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
print('kernel,counter,dispatches,sum,mean_per_dispatch,sum_per_update')
for k in sorted(acc):
    for c in sorted(acc[k]):
        print(f'{k},{c},{cnt[k][c]},{acc[k][c]:.0f},{acc[k][c]/cnt[k][c]:.1f},{acc[k][c]/steps:.1f}')
