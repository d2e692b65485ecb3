cd ${GRAFT_REPO_ROOT:-/root/repo}
WIN=7.3,9.1 bash tools/util_timeline.sh > /dev/null; cp gpurun_out/util_timeline.txt gpurun_out/r06_util_timeline.txt
timeout 300 python tools/gcn_tail_bench.py 32 2>/dev/null | tail -3 > gpurun_out/r06_gcn_tail_bench.txt
timeout 600 python tools/glue_sites.py > gpurun_out/glue_sites.txt 2>&1
timeout 600 python tools/glue_shapes.py > gpurun_out/glue_shapes.txt 2>&1
grep "aten ops" gpurun_out/glue_sites.txt gpurun_out/glue_shapes.txt
