cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "narrow_group or grouped_partial" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 600 python tools/layer_profile.py 2>&1 | grep -E "conv_g3" | head -3
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'])"; }
for i in 1 2; do echo -n "g3    "; b; echo -n "no g3 "; DANET_NO_CONV_G3=1 b; done
