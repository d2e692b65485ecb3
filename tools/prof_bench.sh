#!/bin/bash
# rocprofv3 kernel-trace of bench.py (same command the driver runs); stats CSV -> gpurun_out/
set -u
exec < /dev/null
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-bench}
shift || true
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $REPO/bench.py "$@" > /tmp/prof_bench.log 2>&1
grep -E '^\{' /tmp/prof_bench.log | tail -1 > $REPO/gpurun_out/${TAG}_line.json
cat $REPO/gpurun_out/${TAG}_line.json
for f in $(find /tmp/prof_bench -name "*kernel_stats.csv"); do cp "$f" $REPO/gpurun_out/${TAG}_kernel_stats.csv; done
tail -3 /tmp/prof_bench.log | cut -c1-300
