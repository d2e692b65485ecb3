cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/prof_bench.sh r06_fp32 --dtype fp32 --steps 10 | tail -1 | cut -c1-100
cat gpurun_out/r06_fp32_line.json | cut -c1-400
