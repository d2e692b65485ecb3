#!/bin/bash
# Build the library (so that the in-tree .so the snapshot carries is current), then run a command on the GPU box.
# usage: tools/gpu.sh <timeout seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python danet-densepose2smpl_amd/csrc/build.py > /tmp/danet_build.log 2>&1 || { tail -30 /tmp/danet_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
