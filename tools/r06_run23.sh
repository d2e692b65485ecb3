#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zz_paths.py -m gpu -q -k "regroup or pack_image" 2>&1 | grep -E "Error|err |passed|failed" | head -20
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; }
b on
DANET_REGROUP_PARTS=0 b regroup_off
DANET_PACK_IMAGE=0 b pack_off
b on
DANET_REGROUP_PARTS=0 DANET_PACK_IMAGE=0 b both_off
