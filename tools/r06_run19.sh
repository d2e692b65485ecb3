#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in 1 0 1 0; do DANET_BN_WIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide=$v', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; done
