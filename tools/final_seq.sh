#!/bin/bash
# The round's closing measurements in one gpurun call (fresh box): the full GPU suite THREE times with the tails KEPT
# (profiles/r06_pytest_gpu_{1,2,3}.txt: VERDICT r4 next 1c), the bench line, rocprofv3 of the same command, the bf16-only profile and
# its family breakdown, smoke, the N > 1 code path on a one-rank group, the soak.  tools/pmc_traffic.sh runs after it, as the last GPU action.
exec </dev/null
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=r06
echo "== bench (default flags, fresh box)"
timeout 600 python bench.py > gpurun_out/${R}_final_bench.log 2>&1; grep "^{" gpurun_out/${R}_final_bench.log | tail -1 > gpurun_out/${R}_plain_line.json; cut -c1-300 gpurun_out/${R}_plain_line.json
echo "== rocprofv3 of the same command"
bash tools/prof_bench.sh ${R}_bench | tail -2 | cut -c1-200
echo "== rocprofv3, bf16 steps only"
bash tools/prof_bench.sh ${R}_bf16only --no-fp32 --no-cpu-baseline | tail -1 | cut -c1-100
python tools/step_breakdown.py gpurun_out/${R}_bf16only_kernel_stats.csv > gpurun_out/${R}_step_breakdown.txt; cat gpurun_out/${R}_step_breakdown.txt
rm -f gpurun_out/parity_measured.jsonl
echo "== full GPU suite, three times, tails kept"
for i in 1 2 3; do
  timeout 900 python -m pytest tests -q -x -m gpu > gpurun_out/${R}_pytest_full_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/${R}_pytest_full_$i.log
  # (the summary line first: RCCL prints its banner when the process exits, after pytest's last line)
  (echo "# python -m pytest tests -q -x -m gpu   (run $i of 3, one gpurun box, $(date -u +%FT%TZ))"; grep -h "passed\|failed\|error" gpurun_out/${R}_pytest_full_$i.log | tail -3; tail -4 gpurun_out/${R}_pytest_full_$i.log) > gpurun_out/${R}_pytest_gpu_$i.txt
  tail -2 gpurun_out/${R}_pytest_gpu_$i.txt
done
cp gpurun_out/parity_measured.jsonl gpurun_out/${R}_parity_measured.jsonl 2>/dev/null
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
echo "== the N > 1 code path on a one-rank RCCL group"
timeout 400 python bench.py --force-ddp --no-fp32 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${R}_force_ddp_line.json; cut -c1-200 gpurun_out/${R}_force_ddp_line.json
echo "== soak"
timeout 300 python tools/soak.py 4000 2>&1 | tail -1 > gpurun_out/${R}_soak.txt; cut -c1-1200 gpurun_out/${R}_soak.txt
