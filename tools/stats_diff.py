"""Per-step difference of two rocprofv3 kernel-stats CSVs (bench.py --no-fp32 --no-cpu-baseline runs): which kernels one
configuration launches more of / spends more time in.  usage: python tools/stats_diff.py A.csv B.csv [min_us]  (B - A per step)"""
import csv
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    c = sum(int(r['Calls']) for r in rows if 'conv3x3_stream_kernel<3>' in r['Name'] or 'conv3x3_stream_bn_kernel<3>' in r['Name'])
    steps = c / 132.0 if c else 1.0
    out = {}
    for r in rows:
        if 'Cijk' in r['Name'] and float(r['AverageNs']) > 3e5:
            continue
        out[r['Name']] = (int(r['Calls']) / steps, float(r['TotalDurationNs']) / steps / 1e3)
    return out, steps


def main():
    a, sa = load(sys.argv[1])
    b, sb = load(sys.argv[2])
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
    rows = []
    for k in set(a) | set(b):
        ca, ta = a.get(k, (0, 0.0))
        cb, tb = b.get(k, (0, 0.0))
        rows.append((tb - ta, k, ca, cb, ta, tb))
    rows.sort(reverse=True)
    print('steps: A %.1f  B %.1f   kernel time per step: A %.2f ms  B %.2f ms' % (sa, sb, sum(v[1] for v in a.values()) / 1e3, sum(v[1] for v in b.values()) / 1e3))
    for d, k, ca, cb, ta, tb in rows:
        if abs(d) >= thr:
            print('%+8.1f us  %-90s launches %6.1f -> %6.1f   %8.1f -> %8.1f us' % (d, k[:90], ca, cb, ta, tb))


if __name__ == '__main__':
    main()
