cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { timeout 300 python bench.py --no-cpu-baseline --no-fp32 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
echo -n "plain nv14  "; b
echo -n "plain nv10  "; DANET_LIB=$GRAFT_REPO_ROOT/danet-densepose2smpl_amd/csrc/libdanet_hip_nv10.so b
done
