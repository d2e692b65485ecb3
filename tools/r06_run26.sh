#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4; do echo "== process $i"; timeout 300 python tools/tail_flag_probe.py 2>/dev/null | tail -13; done
