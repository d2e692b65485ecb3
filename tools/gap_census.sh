#!/bin/bash
# rocprofv3 kernel trace of bench.py (graph replays): where the last replay's time goes that is NOT inside a kernel --
# the idle gaps between consecutive dispatches, by the kernel that follows the gap.
set -u
exec < /dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gap_trace
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_trace -o bench -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 "$@" > /tmp/gap_trace.log 2>&1
f=$(find /tmp/gap_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $REPO/gpurun_out/r5_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
step = rows[lo:hi]
S = lambda r: int(r['Start_Timestamp'])
E = lambda r: int(r['End_Timestamp'])
span = (E(step[-1]) - S(step[0])) / 1e6
busy = sum(E(r) - S(r) for r in step) / 1e6
print('dispatches in the last replay: %d   span %.3f ms   sum of kernel durations %.3f ms' % (len(step), span, busy))
gaps, overlap = [], 0.0
end = E(step[0])
by = collections.defaultdict(lambda: [0, 0.0])
for prev, r in zip(step, step[1:]):
    g = S(r) - end
    if g > 0:
        gaps.append(g)
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
        by[k][0] += 1; by[k][1] += g
    else:
        overlap += -g
    end = max(end, E(r))
print('idle between dispatches: %.3f ms in %d gaps (median %.2f us, mean %.2f us); overlapped time %.3f ms' %
      (sum(gaps) / 1e6, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, (sum(gaps) / max(1, len(gaps))) / 1e3, overlap / 1e6))
h = collections.Counter(min(int(g / 1e3), 20) for g in gaps)
print('gap histogram (us: count): ' + ' '.join('%d:%d' % (k, h[k]) for k in sorted(h)))
print('gaps by the kernel that FOLLOWS them:')
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print('  %-60s %4d gaps  %8.1f us  (%.2f us each)' % (k, n, t / 1e3, t / n / 1e3))
PY
