#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps $2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['onepass_error'], [hex(w) for w in d['barrier_error_word']])"; }
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do b window 4; done
b window 20; b window 20
DANET_GCN_TAIL=0 b tail_off 20
