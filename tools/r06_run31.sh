#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['onepass_error'], [hex(w) for w in d['barrier_error_word']])"; }
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do DANET_BODY_HEAD_ON_MAIN=1 b head_on_main; done
for i in 1 2 3 4 5 6; do b default; done
