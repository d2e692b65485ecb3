"""Phase stamps of smpl_lbs_bwd_kernel (workgroup (0,0)): python tools/lbs_phases.py [B]"""
import ctypes
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import assets, _lib                  # noqa: E402
from danet_densepose2smpl_amd.smpl import SMPL                     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
smpl = SMPL(model=assets.make_synthetic_smpl(0)).to(dev)
betas = torch.randn(B, 10, device=dev, requires_grad=True)
rot = torch.eye(3, device=dev).repeat(B, 24, 1, 1).requires_grad_(True)
for it in range(3):
    out = smpl(betas=betas, body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False)
    (out.vertices.sum() + out.joints.sum()).backward()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 16)()
_lib.check(_lib.lib().danet_smpl_lbs_debug(ctypes.addressof(buf)), 'dbg')
st = list(buf)[:7]
names = ['load A/W/Jx', 'seeds', 'd v_posed', 'gA', 'd beta', 'pose-feature chunks']
for i, n in enumerate(names):
    print('%-22s %8d cycles' % (n, st[i + 1] - st[i]))
print('total %d cycles (100 MHz clock64 ticks? see ratio below)' % (st[6] - st[0]))
fw = list(buf)[8:14]
for i, n in enumerate(['forward: load A / W', 'forward: pose blend (incl. feature staging)', 'forward: shape blend', 'forward: skinning', 'forward: extra-joint partials']):
    print('%-46s %8d ticks' % (n, fw[i + 1] - fw[i]))
print('forward total %d ticks' % (fw[5] - fw[0]))

# the one-launch backward (smpl_fused_bwd_kernel, workgroup 0): phase 1 ends at stamp 6, barrier 1 at 7, phase 2 at 14, barrier 2 at 15, the kernel at 8
b = list(buf)
if b[7] > b[6] > 0:
    print('one-launch backward, workgroup 0 (ticks of the 100 MHz clock: 10 ns each):')
    for name, a, z in (('phase 1 (seeds, d v_posed, partials)', 0, 6), ('barrier 1 (incl. waiting for the slowest workgroup)', 6, 7),
                       ('phase 2 (pose-feature contraction)', 7, 14), ('barrier 2', 14, 15), ('phase 3 (reduction + chain)', 15, 8), ('kernel', 0, 8)):
        print('  %-52s %8d' % (name, b[z] - b[a]))
