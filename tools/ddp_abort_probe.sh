#!/bin/bash
# Probe for the sporadic abort of the 1-rank data-parallel graph path: the test N times in fresh processes per variant.
#   gpurun -- 'bash tools/ddp_abort_probe.sh 6 "base nocache sleep nosegments"'
N=${1:-6}
VARIANTS=${2:-base}
mkdir -p gpurun_out
ulimit -c 0
for v in $VARIANTS; do
  bad=0
  for i in $(seq 1 $N); do
    case $v in
      base) EXTRA="" ;;
      nocache) EXTRA="TORCH_NCCL_CUDA_EVENT_CACHE=0" ;;
      sleep) EXTRA="DANET_CAPTURE_SETTLE=0.5" ;;
      nosegments) EXTRA="DANET_SEGMENTS=0" ;;
      *) EXTRA="$v" ;;
    esac
    env $EXTRA TORCH_SHOW_CPP_STACKTRACES=0 timeout 300 python -m pytest tests/test_gpu_models.py -q -x -s \
        -k "test_data_parallel_graph_path_single_rank" > gpurun_out/ddp_probe_${v}_$i.log 2>&1
    rc=$?
    [ $rc -ne 0 ] && bad=$((bad+1))
  done
  echo "variant $v: $bad / $N failed" | tee -a gpurun_out/ddp_probe_summary.txt
done
