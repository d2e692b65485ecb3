cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
b() { timeout 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'], (d.get('fp32') or {}).get('ms_per_step'), (d.get('allreduce') or {}).get('released_during_backward'))"; }
echo -n "plain (with fp32 record) "; b
echo -n "force-ddp                "; b --no-fp32 --force-ddp
echo -n "force-ddp, streams off   "; DANET_BODY_STREAM=0 DANET_HEAD_STREAM=0 b --no-fp32 --force-ddp
