#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gcn_tail.py -m gpu -x -q 2>&1 | tail -15
for v in 1 0 1 0; do DANET_GCN_TAIL=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail=$v', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; done
