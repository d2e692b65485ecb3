"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/prof_top.py <dir> [n]."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', round(tot / 1e6, 2), 'kernels', len(rows), 'launches', sum(int(r['Calls']) for r in rows))
for r in rows[:n]:
    print('%-100s %6s %10.3f ms %9.1f us %5s%%' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
