"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (csv output directory)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print('%-62s calls %6s avg %9.1f us  %5.1f%%' % (r['Name'][:62], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
