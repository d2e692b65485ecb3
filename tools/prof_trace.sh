#!/bin/bash
# rocprofv3 kernel trace of bench.py; per-dispatch durations of the kernels matching $1 (regex) of the LAST step
set -u
exec < /dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
PAT=${1:-wgrad}
shift || true
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trace
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o bench -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > /tmp/prof_trace.log 2>&1
f=$(find /tmp/prof_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$PAT" <<'PY' | tee $REPO/gpurun_out/trace_${PAT//[^a-zA-Z0-9]/_}.txt
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last graph replay = the last occurrence of adam_kernel backwards to the previous one
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
step = rows[lo:hi]
t0 = int(step[0]['Start_Timestamp'])
print('dispatches in the last step:', len(step), ' span ms: %.3f' % ((int(step[-1]['End_Timestamp']) - t0) / 1e6))
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print('sum of kernel durations ms: %.3f' % (busy / 1e6))
for r in step:
    if pat.search(r['Kernel_Name']):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print('%9.1f us  @%8.3f ms  grid %-8s %s' % (d, (int(r['Start_Timestamp']) - t0) / 1e6, r.get('Grid_Size_X', '?'), r['Kernel_Name'][:90]))
PY
