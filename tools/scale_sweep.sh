#!/bin/bash
# First node-hour on an 8-GPU MI355X node: one curve and one setting instead of a debugging session.
# For N in 1 2 4 8, communication-channel limits 8 / 16 / 24 / 32 (DANET_COMM_CHANNELS -> NCCL_MAX_NCHANNELS, which also sizes
# the one-pass BatchNorm barrier: 2 * (256 - channels) workgroups) and both wire formats, the bench line of every run is kept
# under gpurun_out/scale_sweep/ and a summary table is printed.  What to read, in order ("first real run"):
#   1. every line: "finite_losses_and_parameters": true, "onepass_error": false, allreduce.poison_sum 0.0, allreduce.mode "in-graph"
#      (mode "after the graph replay" = the capture with collectives failed and the fallback ran: note the stderr of that run);
#   2. allreduce.ms_per_step_per_rank: a straggler rank shows here before it shows in the aggregate;
#   3. value(N) / (N * value(1)) per (channels, wire): the scaling efficiency; pick the channel count where it peaks.
# usage: tools/scale_sweep.sh [steps] [gpu counts] ; needs the node's GPUs visible to this shell.
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-30}
NS=${2:-"1 2 4 8"}
OUT=gpurun_out/scale_sweep
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for wire in fp32 bf16; do
  for ch in 8 16 24 32; do
    for n in $NS; do
      tag=n${n}_ch${ch}_${wire}
      port=$((29500 + RANDOM % 2000))
      if [ "$n" = 1 ]; then
        DANET_COMM_CHANNELS=$ch timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-fp32 --no-cpu-baseline --force-ddp --grad-wire $wire \
          > $OUT/$tag.log 2> $OUT/$tag.err
      else
        DANET_COMM_CHANNELS=$ch timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
          bench.py --gpus $n --steps $STEPS --warmup 5 --no-fp32 --no-cpu-baseline --grad-wire $wire > $OUT/$tag.log 2> $OUT/$tag.err
      fi
      grep '^{' $OUT/$tag.log | tail -1 > $OUT/$tag.json
    done
  done
done
python - <<'PY'
import glob, json, os
rows = {}
for f in sorted(glob.glob('gpurun_out/scale_sweep/*.json')):
    try:
        d = json.load(open(f))
    except Exception:
        print(os.path.basename(f), 'NO LINE (see the .err file)'); continue
    tag = os.path.basename(f)[:-5]
    n, ch, wire = tag.split('_')
    ar = d.get('allreduce') or {}
    rows[(wire, ch, int(n[1:]))] = (d['value'], d['ms_per_step'], ar.get('mode'), d.get('finite_losses_and_parameters'), d.get('onepass_error'), ar.get('poison_sum'),
                                    max(ar.get('ms_per_step_per_rank') or [0]) - min(ar.get('ms_per_step_per_rank') or [0]))
print('%-5s %-5s %3s %10s %9s %6s  %-24s %s' % ('wire', 'chan', 'N', 'img/s', 'ms/step', 'eff', 'all-reduce mode', 'finite / onepass_error / poison_sum / rank spread ms'))
for (wire, ch, n), v in sorted(rows.items()):
    base = rows.get((wire, ch, 1))
    eff = '%.3f' % (v[0] / (n * base[0])) if base else '-'
    print('%-5s %-5s %3d %10.1f %9.3f %6s  %-24s %s / %s / %s / %.3f' % (wire, ch, n, v[0], v[1], eff, v[2], v[3], v[4], v[5], v[6]))
PY
