#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do echo "== $i"; DANET_BENCH_STAGES=1 timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 6 --warmup 2 2>&1 >/dev/null | grep "^stage" | awk '{print $2,$3,$4,$5,$6,$7}'; done
