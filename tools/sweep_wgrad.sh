#!/bin/bash
exec </dev/null
for t in 64 128 256 512 1024 2048; do
  echo "blocks=$t"; DANET_WGRAD_BLOCKS=$t timeout 100 python tools/microbench_conv.py 32 nomiopen 2>&1 | grep shape | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['shape'], 'wgrad_us', d['wgrad_us'])
"
done
