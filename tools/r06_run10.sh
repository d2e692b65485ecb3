cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "pair_mode or deferred_multi or wgrad3x3_direct or transpose_read" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'])"; }
for i in 1 2; do echo -n "small-map "; b; echo -n "no pair   "; DANET_NO_WGRAD3_PAIR=1 b; done
