#!/bin/bash
# rocprofv3 kernel trace of bench.py -> a timeline of the LAST step in 0.25 ms buckets: how much of each bucket has a kernel running at
# all (union of intervals), how well those kernels fill the chip (workgroups / 256 CUs, capped at 1, time-weighted), and which kernels
# sit in the under-filled buckets.  Finds the launch-bound stretches a side stream could hide.
set -u
exec < /dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o bench -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 "$@" > /tmp/prof_tl.log 2>&1
f=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $REPO/gpurun_out/util_timeline.txt
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
step = rows[lo:hi]
t0 = int(step[0]['Start_Timestamp'])
T = int(step[-1]['End_Timestamp']) - t0
B = 250000
nb = T // B + 1
busy = [0.0] * nb; fill = [0.0] * nb; names = [collections.Counter() for _ in range(nb)]
def wgs(r):
    g = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    w = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    return max(1, g // max(1, w))
ev = []
def nm(r):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    return re.split(r'[(<]', n)[0][-40:] + ':%d' % wgs(r)
for r in step:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    f = min(1.0, wgs(r) / 256.0)
    b = s // B
    while b * B < e:
        a, z = max(s, b * B), min(e, (b + 1) * B)
        fill[b] += f * (z - a); names[b][nm(r)] += z - a
        b += 1
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = 0
for t, d in ev:
    if depth > 0:
        b = last // B
        while b * B < t:
            a, z = max(last, b * B), min(t, (b + 1) * B)
            busy[b] += z - a; b += 1
    depth += d; last = t
print('last step: %d dispatches, span %.3f ms' % (len(step), T / 1e6))
under = 0.0
for b in range(nb):
    fl = min(1.0, fill[b] / B)
    top = ', '.join('%s %.0f' % (k, v / 1e3) for k, v in names[b].most_common(3))
    mark = ' <<' if fl < 0.35 else ''
    under += (1 - min(1.0, fl)) * B if fl < 0.35 else 0
    print('%6.2f ms  busy %.2f  fill %.2f%s   %s' % (b * B / 1e6, busy[b] / B, fl, mark, top))
import os
if os.environ.get('WIN'):
    lo_ms, hi_ms = [float(x) for x in os.environ['WIN'].split(',')]
    prev = None
    print('--- dispatches starting in [%.2f, %.2f) ms: start, duration us, gap to previous end us, workgroups, name' % (lo_ms, hi_ms))
    for r in step:
        s_, e_ = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        if lo_ms * 1e6 <= s_ < hi_ms * 1e6:
            print('%8.3f %7.1f %6.1f %6d  %s' % (s_ / 1e6, (e_ - s_) / 1e3, (s_ - prev) / 1e3 if prev is not None else 0, wgs(r), nm(r)))
        prev = e_ if prev is None else max(prev, e_)
print('time in buckets with fill < 0.35: %.2f ms' % (sum(1 for b in range(nb) if min(1.0, fill[b] / B) < 0.35) * B / 1e6))
PY
