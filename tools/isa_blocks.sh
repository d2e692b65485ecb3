#!/bin/bash
# Static view of a kernel without a GPU: registers, code size, and the instruction mix + waitcnts of every basic block that holds
# MFMAs.  usage: tools/isa_blocks.sh <csrc file> <mangled-name substring> [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1; K=$2; shift 2
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form --offload-device-only -S -I$R/include \
  -I$R/danet-densepose2smpl_amd/csrc "$@" $R/danet-densepose2smpl_amd/csrc/$F -o /tmp/isa.s 2>/dev/null || exit 1
python3 - "$K" <<'PY'
import re, sys
key = sys.argv[1]
txt = open('/tmp/isa.s').read()
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', txt, re.M) if key in m.group(1)]
for name in names:
    a = txt.index('\n' + name + ':')
    b = txt.index('s_endpgm', a)
    body = txt[a:b]
    meta = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n.*?\.end_amdhsa_kernel', txt, re.S)
    vg = re.search(r'next_free_vgpr (\d+)', meta.group(0)).group(1) if meta else '?'
    print('==', name[:110], 'vgprs', vg)
    blocks, cur, bn = [], [], 'entry'
    for l in body.split('\n'):
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append((bn, cur)); bn = l.split(':')[0]; cur = []
        else:
            cur.append(l)
    blocks.append((bn, cur))
    for n, bl in blocks:
        ins = [x.strip().split()[0] for x in bl if x.startswith('\t') and x.strip() and not x.strip().startswith(('.', ';'))]
        m = sum(i.startswith('v_mfma') for i in ins)
        if m >= 4:
            print('  ', n, len(ins), 'mfma', m, 'ds', sum(i.startswith('ds_') for i in ins), 'vmem', sum(i.startswith(('buffer', 'global', 'flat')) for i in ins),
                  'valu', sum(i.startswith('v_') and not i.startswith('v_mfma') for i in ins), 'salu', sum(i.startswith('s_') and not i.startswith('s_waitcnt') for i in ins),
                  'waits', [x.strip().replace('s_waitcnt ', '') for x in bl if 's_waitcnt' in x][:16])
PY
