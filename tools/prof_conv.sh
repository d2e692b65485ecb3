#!/bin/bash
# rocprofv3 kernel trace of tools/microbench_conv.py; per (kernel, grid) average / min duration
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_conv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_conv -o c -- python $REPO/tools/microbench_conv.py ${1:-32} nomiopen > /tmp/prof_conv.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_conv/*kernel_trace.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'conv' not in n:
        continue
    grid = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
    agg[(n, grid, r.get('LDS_Block_Size'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, grid, lds), v in sorted(agg.items(), key=lambda kv: kv[0]):
    if len(v) >= 20:
        v.sort()
        print('%-44s grid=%-22s lds=%-7s n=%-4d med=%7.1f us min=%7.1f' % (n[:44], ','.join(map(str, grid)), lds, len(v), v[len(v) // 2], v[0]))
PY
