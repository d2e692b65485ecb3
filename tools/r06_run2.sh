cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
echo "== stn bench"; timeout 300 python tools/stn_bench.py 2>&1 | tail -2
echo "== targeted tests"; timeout 900 python -m pytest tests/test_gpu_f2.py tests/test_gpu_fp32.py tests/test_gpu_norm.py tests/test_gpu_parts.py -x -q 2>&1 | tail -8
cat gpurun_out/parity_measured.jsonl
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 > gpurun_out/r06_b1.log 2>&1; grep "^{" gpurun_out/r06_b1.log | tail -1 > gpurun_out/r06_b1_line.json; cut -c1-300 gpurun_out/r06_b1_line.json
echo "== full suite"; timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r06_pytest_1.log 2>&1; tail -5 gpurun_out/r06_pytest_1.log
