cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_iuv.py tests/test_gpu_parts.py tests/test_gpu_zz_paths.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'])"; }
for i in 1 2; do echo -n "finalize   "; b; echo -n "tensor ops "; DANET_LOSS_FINALIZE=0 b; done
