cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_norm.py -x -q -k "channel_sum" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'])"; }
for i in 1 2; do echo -n "16 copies "; b; echo -n "1 copy    "; DANET_CHSUM_COPIES=1 b; done
