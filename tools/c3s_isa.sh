#!/bin/bash
# Static view of the streamed 3x3 kernel (csrc/conv3x3s.hip) without a GPU: registers, code size, and the instruction mix of every
# basic block that holds MFMAs (the k-loop's group block is the one with D * 12 of them).  usage: tools/c3s_isa.sh [NT] [extra hipcc flags]
NT=${1:-3}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --offload-device-only -S -I$R/include \
  -I$R/danet-densepose2smpl_amd/csrc "$@" ${SRC:-$R/danet-densepose2smpl_amd/csrc/conv3x3s.hip} -o /tmp/c3s.s 2>/dev/null || exit 1
awk "/^_ZN12_GLOBAL__N_121conv3x3_stream_kernelILi${NT}EEEvNS_8S3LaunchE:/,/s_endpgm/" /tmp/c3s.s > /tmp/k3.s
awk "/^_ZN12_GLOBAL__N_121conv3x3_stream_kernelILi${NT}EEEvNS_8S3LaunchE:/{f=1} f&&/codeLenInByte|NumVgprs|sgpr_spill|ScratchSize/{print} f&&/Occupancy/{exit}" /tmp/c3s.s
python3 - <<'PY'
import re
lines = open('/tmp/k3.s').read().split('\n')
blocks, cur, name = [], [], 'entry'
for l in lines:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.split(':')[0]; cur = []
    else:
        cur.append(l)
blocks.append((name, cur))
tot = 0
for n, b in blocks:
    ins = [x.strip().split()[0] for x in b if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    tot += len(ins)
    m = sum(i.startswith('v_mfma') for i in ins)
    if m >= 12:
        print(n, len(ins), 'mfma', m, 'ds_read', sum(i.startswith('ds_read') for i in ins), 'buffer', sum(i.startswith('buffer') for i in ins),
              'valu', sum(i.startswith('v_') and not i.startswith('v_mfma') for i in ins), 'salu', sum(i.startswith('s_') for i in ins),
              'waitcnt', [x.strip() for x in b if 's_waitcnt' in x][:12])
print('instructions', tot)
PY
