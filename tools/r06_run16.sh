cd $GRAFT_REPO_ROOT
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'])"; }
for i in 1 2; do echo -n "part loss on side "; b; echo -n "part loss on main "; DANET_PART_LOSS_STREAM=0 b; done
