#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
b() { s=$(date +%s.%N); timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['finite_losses_and_parameters'], d['onepass_error'], end=' ')"; e=$(date +%s.%N); echo "wall $(echo "$e - $s" | bc)"; }
timeout 600 python -m pytest tests/test_gpu_zz_paths.py -m gpu -q -k "regroup or pack_image" 2>&1 | tail -1
for i in 1 2 3 4 5 6 7 8; do b tail_on; DANET_GCN_TAIL=0 b tail_off; done
