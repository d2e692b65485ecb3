#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_bench.sh r06_tmp --no-fp32 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_tmp_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    print('%-70s calls %5s avg %8.1f us  total/step %7.3f ms' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/30/1e6))
PY
