#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; }
for i in 1 2 3 4 5 6; do b all_on; done
for i in 1 2 3 4; do DANET_GCN_TAIL=0 b tail_off; done
for i in 1 2 3 4; do DANET_BN_WIDE=0 b wide_off; done
