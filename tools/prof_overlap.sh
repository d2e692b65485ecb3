#!/bin/bash
# kernel trace of a short bench run: how much do kernels overlap inside the hipGraph replay?
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ov
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -o b -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/prof_ov.log 2>&1
tail -1 /tmp/prof_ov.log | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_ov/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id'), r.get('Stream_Id')) for r in csv.DictReader(open(f))]
rows.sort()
# the last replay = the last ~N kernels; take the final 4000 kernels
tail = rows[-4000:]
t0, t1 = tail[0][0], max(r[1] for r in tail)
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in tail:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in tail)
print('last 4000 kernels: wall %.2f ms, union-busy %.2f ms, sum of durations %.2f ms, idle %.2f ms' % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, (t1 - t0 - busy) / 1e6))
import collections
q = collections.Counter((r[3], r[4]) for r in tail)
print('queues/streams:', dict(q))
gaps = sorted([tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)])
print('gap between consecutive kernels (start - prev end), ns: median %d, p10 %d, p90 %d' % (gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[len(gaps) * 9 // 10]))
PY
