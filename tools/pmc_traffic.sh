#!/bin/bash
# HBM traffic of the step's kernels from the PMC counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (they do not fit one), kernel-trace only, calibrated on a copy of known size in
# the same run (on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x; other widths are uncalibrated).
# Output: gpurun_out/pmc_traffic.json  {kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch}}
exec </dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  DANET_PMC_CALIB=1 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o c -- python $REPO/tools/run_step.py 32 256 2 > /tmp/pmc_$C.log 2>&1
  tail -1 /tmp/pmc_$C.log | cut -c1-160
done
python - <<'PY'
import csv, glob, json, collections, os
out = {}
calib = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(list)
    for f in glob.glob('/tmp/pmc_%s/*counter_collection.csv' % C):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != C:
                continue
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            agg[n].append((float(r['Counter_Value']), int(r.get('Dispatch_Id') or 0)))
    # calibration: the 512 MiB bf16 copy (read 512 MiB, write 512 MiB) is the LAST copy kernel of the run
    best = None
    for n, v in agg.items():
        if 'copy' in n.lower():
            for val, did in v:
                if best is None or did > best[1]:
                    best = (val, did, n)
    calib[C] = {'counter_value': best[0], 'kernel': best[2], 'true_bytes': 512 * 2**20, 'bytes_per_count': 512 * 2**20 / best[0]}
    for n, v in agg.items():
        if 'conv' in n or 'bn_' in n or 'wgrad' in n:
            d = out.setdefault(n, {})
            d['launches'] = len(v)
            d[('fetch' if C == 'FETCH_SIZE' else 'write') + '_bytes_per_launch'] = sum(x for x, _ in v) / len(v) * calib[C]['bytes_per_count']
json.dump({'calibration': calib, 'kernels': out}, open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'pmc_traffic.json'), 'w'), indent=1)
for n, d in sorted(out.items(), key=lambda kv: -kv[1].get('fetch_bytes_per_launch', 0) * kv[1]['launches'])[:14]:
    print('%-44s n=%-5d fetch %.2f MB  write %.2f MB per launch' % (n[:44], d['launches'], d.get('fetch_bytes_per_launch', 0) / 1e6, d.get('write_bytes_per_launch', 0) / 1e6))
print(calib)
PY
