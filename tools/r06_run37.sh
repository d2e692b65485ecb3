#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "== whole-step soak"
timeout 600 python tools/step_soak.py 6000 2>/dev/null | tail -1 > gpurun_out/r06_step_soak.json; cut -c1-400 gpurun_out/r06_step_soak.json
echo "== util timeline"
WIN=7.5,9.3 bash tools/util_timeline.sh > /dev/null; cp gpurun_out/util_timeline.txt gpurun_out/r06_util_timeline.txt; tail -3 gpurun_out/util_timeline.txt
echo "== gcn tail bench"
timeout 300 python tools/gcn_tail_bench.py 32 2>/dev/null | tail -3 > gpurun_out/r06_gcn_tail_bench.txt; cat gpurun_out/r06_gcn_tail_bench.txt
echo "== pmc traffic"
bash tools/pmc_traffic.sh 2>&1 | tail -3
