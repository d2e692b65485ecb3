#!/bin/bash
# PMC instruction mix of every kernel of a few eager train steps (own pass; kernel-trace only)
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_step
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/pmc_step -o c -- python $REPO/tools/run_step.py 32 256 3 > /tmp/pmc_step.log 2>&1
tail -2 /tmp/pmc_step.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob('/tmp/pmc_step/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')[:70]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES':
            cnt[n] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))
tot = sum(v.get('SQ_WAVE_CYCLES', 0) for _, v in rows)
print('%-70s %6s %6s %9s %7s %7s %6s %6s %6s %7s' % ('kernel', 'calls', '%wcyc', 'waves/call', 'VALU/w', 'SALU/w', 'VMRD/w', 'VMWR/w', 'LDS/w', 'MFMA/w'))
for n, v in rows[:45]:
    w = max(v.get('SQ_WAVES', 1), 1)
    print('%-70s %6d %6.2f %9.0f %7.0f %7.0f %6.0f %6.0f %6.0f %7.0f' % (n, cnt[n], 100 * v.get('SQ_WAVE_CYCLES', 0) / tot, w / max(cnt[n], 1), v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_SALU', 0) / w,
          v.get('SQ_INSTS_VMEM_RD', 0) / w, v.get('SQ_INSTS_VMEM_WR', 0) / w, v.get('SQ_INSTS_LDS', 0) / w, v.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) / w / 32))
PY
