#!/bin/bash
# PMC counters (own pass, kernel-trace only) for the conv microbench: instruction mix per wave
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_conv
timeout 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16} --output-format csv -d /tmp/pmc_conv -o c -- python $REPO/tools/microbench_conv.py ${1:-32} nomiopen > /tmp/pmc_conv.log 2>&1
tail -2 /tmp/pmc_conv.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/pmc_conv/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'conv' not in n and 'bn_' not in n:
            continue
        agg[(n, r.get('Grid_Size'))][r['Counter_Name']].append(float(r['Counter_Value']))
for (n, g), cs in sorted(agg.items()):
    w = sum(cs.get('SQ_WAVES', [0])) / max(1, len(cs.get('SQ_WAVES', [1])))
    line = '%-40s grid=%-8s waves=%-6d' % (n[:40], g, w)
    for c, v in sorted(cs.items()):
        if c != 'SQ_WAVES':
            line += ' %s/wave=%.0f' % (c.replace('SQ_', ''), (sum(v) / len(v)) / max(w, 1))
    print(line)
PY
