#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['onepass_error'], [hex(w) for w in d['barrier_error_word']])"; }
b cap; DANET_SIDE_CAP=0 b nocap; b cap; DANET_SIDE_CAP=0 b nocap
