cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== stn bench"; timeout 300 python tools/stn_bench.py 2>&1 | grep "^\[" ; DANET_STN_V1=1 timeout 300 python tools/stn_bench.py 2>&1 | grep "^\["
echo "== A-B bench"; for v in 0 1 0 1; do DANET_STN_V1=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('STN_V1=$v', d['ms_per_step'])"; done
