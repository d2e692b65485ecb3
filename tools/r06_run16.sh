cd $GRAFT_REPO_ROOT
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; }
for i in 1 2; do echo -n "heads on side stream "; b; echo -n "heads on main        "; DANET_HEAD_STREAM=0 b; done
timeout 900 python -m pytest tests/test_gpu_zz_paths.py tests/test_gpu_models.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
