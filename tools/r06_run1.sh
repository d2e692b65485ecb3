cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_f2.py "tests/test_gpu_fp32.py::test_hrnet_fp32_vs_reference_golden_at_the_benched_resolution" "tests/test_gpu_models.py::test_hrnet_stages_teacher_forced_vs_fp32_oracle" tests/test_gpu_smpl.py -x -q 2>&1 | tail -15
cat gpurun_out/parity_measured.jsonl
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06_b0.log 2>&1; grep "^{" gpurun_out/r06_b0.log | tail -1 > gpurun_out/r06_b0_line.json; cut -c1-400 gpurun_out/r06_b0_line.json
echo "== full suite"; timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r06_pytest_0.log 2>&1; tail -5 gpurun_out/r06_pytest_0.log
