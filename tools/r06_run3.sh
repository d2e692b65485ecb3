cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== stn bench"; timeout 300 python tools/stn_bench.py 2>&1 | grep "^\[" ; DANET_STN_V1=1 timeout 300 python tools/stn_bench.py 2>&1 | grep "^\["
timeout 300 python -m pytest tests/test_gpu_norm.py -q -x -k stn 2>&1 | tail -2
