cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_layers.py tests/test_gpu_zz_paths.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'])"; }
for i in 1 2; do echo -n "subsets "; b; echo -n "singles "; DANET_MULTI_DGRAD_SUBSETS=0 b; done
