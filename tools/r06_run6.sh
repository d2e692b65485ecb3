cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo -n "default (npm 20, 768)  "; b
for npm in 6 10; do for blk in 768 1152 1536; do echo -n "npm $npm blocks $blk   "; DANET_WGRAD3_NPM=$npm DANET_WGRAD3_MULTI_BLOCKS=$blk b; done; done
for blk in 1152 1536; do echo -n "npm 20 blocks $blk   "; DANET_WGRAD3_MULTI_BLOCKS=$blk b; done
echo -n "default (npm 20, 768)  "; b
