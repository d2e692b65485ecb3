cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/prof_bench.sh r06_plain --no-fp32 --no-cpu-baseline | tail -1 | cut -c1-120
bash tools/prof_bench.sh r06_ddp --no-fp32 --no-cpu-baseline --force-ddp | tail -1 | cut -c1-120
python tools/step_breakdown.py gpurun_out/r06_plain_kernel_stats.csv > gpurun_out/r06_plain_breakdown.txt
python tools/step_breakdown.py gpurun_out/r06_ddp_kernel_stats.csv > gpurun_out/r06_ddp_breakdown.txt
python tools/stats_diff.py gpurun_out/r06_plain_kernel_stats.csv gpurun_out/r06_ddp_kernel_stats.csv 10 > gpurun_out/r06_ddp_diff.txt
cat gpurun_out/r06_ddp_diff.txt | head -50
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-fp32 --force-ddp 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force-ddp', d['ms_per_step'], d['allreduce'])"; done
