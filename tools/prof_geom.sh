#!/bin/bash
# rocprofv3 kernel-trace of the SMPL/raster micro-benchmark; writes a stats CSV into gpurun_out/
set -u
exec < /dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_geom
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geom -o geom -- python $REPO/tools/microbench_geom.py ${1:-32} > /tmp/prof_geom.log 2>&1
grep -E '^\{' /tmp/prof_geom.log
for f in $(find /tmp/prof_geom -name "*kernel_stats.csv"); do cp "$f" $REPO/gpurun_out/${2:-geom}_kernel_stats.csv; done
ls /tmp/prof_geom/* | head
head -12 $REPO/gpurun_out/${2:-geom}_kernel_stats.csv 2>/dev/null | cut -c1-200
