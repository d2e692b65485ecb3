#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gcn_tail.py -m gpu -q 2>&1 | grep -E "Error|err |passed|failed" | head -20
WIN=7.5,10.5 bash tools/util_timeline.sh > /dev/null; grep -E "gcn_tail|smpl_fused_fwd|rot6d" gpurun_out/util_timeline.txt | head
