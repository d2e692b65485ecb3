"""The fused graph tail (csrc/gcn_tail.hip) alone at the benched batch: launch durations (events) and the phase stamps of workgroup 0."""
import ctypes
import sys

import torch

sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tests')
from test_gpu_gcn_tail import _net, _cfg, _tail          # noqa: E402
from danet_densepose2smpl_amd import _lib                # noqa: E402

_cfg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = _net(1)
x = (torch.randn(B, 24, 128).abs() * 0.7).cuda().requires_grad_(True)
L = _lib.lib()
for it in range(3):
    net.zero_grad(set_to_none=True)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    o = _tail(net, x)
    e[1].record()
    loss = sum(t.sum() for t in o)
    torch.cuda.synchronize()
    e2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e2[0].record()
    loss.backward()
    e2[1].record()
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 32)()
    L.danet_gcn_tail_debug(st)
    f = [(st[i] - st[0]) / 100.0 for i in range(11)]
    b = [(st[i] - st[16]) / 100.0 for i in range(16, 29)]
    print('iter %d: forward %.1f us, backward %.1f us (host-side events, include launch overheads)' % (it, e[0].elapsed_time(e[1]) * 1e3, e2[0].elapsed_time(e2[1]) * 1e3))
    print('  fwd stamps us:', ' '.join('%.1f' % v for v in f))
    print('  bwd stamps us:', ' '.join('%.1f' % v for v in b))
