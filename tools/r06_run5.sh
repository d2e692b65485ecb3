cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); a=d.get('allreduce') or {}; print(d['ms_per_step'], a.get('released_during_backward'), d['finite_losses_and_parameters'], d['onepass_error'])"; }
for i in 1 2; do
echo -n "plain            "; b
echo -n "ddp grouped      "; b --force-ddp
done
timeout 900 python -m pytest tests/test_gpu_norm.py tests/test_gpu_zz_paths.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
