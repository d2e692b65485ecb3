#!/bin/bash
exec </dev/null
for a in 0 1 2 3 4 8 7 15; do
  echo -n "ablate=$a  "; DANET_WGRAD_ABLATE=$a DANET_WGRAD_BLOCKS=${1:-384} timeout 100 python tools/microbench_conv.py 32 nomiopen 2>&1 | grep shape | head -1 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], 'wgrad_us', d['wgrad_us'])
"
done
