#!/bin/bash
# PMC counters (own passes, kernel-trace only) of the stem kernels (tools/stem_bench.py): instruction mix, MFMA busy cycles, waits, LDS bank
# conflicts -- per wave.  Output: gpurun_out/pmc_stem.txt
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $REPO/gpurun_out/pmc_stem.txt
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_stem
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_stem -o c -- python $REPO/tools/stem_bench.py > /tmp/pmc_stem.log 2>&1
  python - "$SET" <<'PY' >> $REPO/gpurun_out/pmc_stem.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_stem/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'conv_stem' not in n:
            continue
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
print('# ' + sys.argv[1])
for n, cs in sorted(agg.items()):
    w = sum(cs.get('SQ_WAVES', [0])) / max(1, len(cs.get('SQ_WAVES', [1])))
    line = '%-26s waves=%-5d' % (n[:26], w)
    for c, v in sorted(cs.items()):
        if c != 'SQ_WAVES':
            line += ' %s/wave=%.0f' % (c.replace('SQ_', ''), (sum(v) / len(v)) / max(w, 1))
    print(line)
PY
done
cat $REPO/gpurun_out/pmc_stem.txt
