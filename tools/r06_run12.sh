cd $GRAFT_REPO_ROOT; rm -f gpurun_out/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_fp32.py -x -q -k "benched" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
cat gpurun_out/parity_measured.jsonl | cut -c1-1500
