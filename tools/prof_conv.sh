#!/bin/bash
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_conv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_conv -o c -- python $REPO/tools/microbench_conv.py ${1:-32} nomiopen > /tmp/prof_conv.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_conv/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if float(r['Percentage']) > 0.3:
        print('%6.2f%% calls=%-5s avg=%8.1f us min=%8.1f  %s' % (float(r['Percentage']), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, n[:70]))
PY
