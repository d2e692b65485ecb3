#!/bin/bash
exec </dev/null
for t in 128 256 384 512 1024; do
  echo "blocks=$t"; DANET_WGRAD3_BLOCKS=$t timeout 100 python tools/microbench_conv.py 32 nomiopen 2>&1 | grep shape | head -6 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['shape'], 'wgrad3_us', d.get('wgrad3_us'))
"
done
