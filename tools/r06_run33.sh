#!/bin/bash
exec < /dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
rm -rf /tmp/prof_to
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_to -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 > /tmp/prof_to.log 2>&1
f=$(find /tmp/prof_to -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
long = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 50_000_000 and 'Cijk' not in r['Kernel_Name']]
print('run: %d dispatches, %d kernels longer than 50 ms' % (len(rows), len(long)))
for r in long[:3]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('LONG %.1f ms  %s grid %s queue %s' % ((e - s) / 1e6, r['Kernel_Name'][:60], r['Grid_Size_X'], r.get('Queue_Id')))
    # everything that overlaps it
    for q in rows:
        qs, qe = int(q['Start_Timestamp']), int(q['End_Timestamp'])
        if q is not r and qs < e and qe > s and (qe - qs > 1_000_000 or qs < s):
            print('    overlaps: %.3f ms (start %+.3f ms) %s grid %s wg %s queue %s' % ((qe - qs) / 1e6, (qs - s) / 1e6, q['Kernel_Name'][:70], q['Grid_Size_X'], q['Workgroup_Size_X'], q.get('Queue_Id')))
PY
done
