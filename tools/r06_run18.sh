#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_norm.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; done
DANET_BODY_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-fp32 --force-ddp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-ddp', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"
