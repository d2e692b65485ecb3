#!/bin/bash
# The round's closing measurements in one gpurun call (fresh box): the bench line, rocprofv3 of the same command, the bf16-only profile and
# its family breakdown, the full GPU suite, the soak.  tools/pmc_traffic.sh runs after it, as the last GPU action.
exec </dev/null
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== bench (default flags, fresh box)"
timeout 600 python bench.py > gpurun_out/r04_final_bench.log 2>&1; grep "^{" gpurun_out/r04_final_bench.log | tail -1 > gpurun_out/r04_plain_line.json; cut -c1-400 gpurun_out/r04_plain_line.json
echo "== rocprofv3 of the same command"
bash tools/prof_bench.sh r04_bench | tail -2 | cut -c1-200
echo "== rocprofv3, bf16 steps only"
bash tools/prof_bench.sh r04_bf16only --no-fp32 --no-cpu-baseline | tail -1 | cut -c1-100
python tools/step_breakdown.py gpurun_out/r04_bf16only_kernel_stats.csv > gpurun_out/r04_step_breakdown.txt; cat gpurun_out/r04_step_breakdown.txt
echo "== full GPU suite"
timeout 1300 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head -8
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
echo "== soak"
timeout 300 python tools/soak.py 4000 2>&1 | tail -1 | cut -c1-900
