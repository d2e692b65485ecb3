cd $GRAFT_REPO_ROOT
timeout 200 python tools/wgrad3_bench.py 2>&1 | tail -1
DANET_WGRAD3_XCD=0 timeout 200 python tools/wgrad3_bench.py 2>&1 | tail -1
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do echo -n "xcd order "; b; echo -n "plain     "; DANET_WGRAD3_XCD=0 b; done
