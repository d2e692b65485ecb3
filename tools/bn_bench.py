"""Timing of the multi-tensor BatchNorm forward+backward on the four HRNet branch shapes: one-pass vs two-kernel backward."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import nn as dnn   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(B, 48, 64, 64), (B, 96, 32, 32), (B, 192, 16, 16), (B, 384, 8, 8)]
bns = [dnn.BatchNorm2d(s[1]).cuda().train() for s in shapes]
xs = [torch.randn(s, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) for s in shapes]
gs = [torch.randn(s, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) for s in shapes]


def run(res, n=1):
    xa = [x.clone().requires_grad_(True) for x in xs[:n]]
    ra = [x.clone().requires_grad_(True) for x in xs[:n]] if res else [None] * n
    ya = dnn.multi_batch_norm(bns[:n], xa, ra, relu=True)
    torch.autograd.backward(ya, gs[:n])


for n in (4, 1):
    for res in (False, True):
        for one in (True, False):
            dnn.ONEPASS = one
            for _ in range(5):
                run(res, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run(res, n)
            e1.record()
            torch.cuda.synchronize()
            print('branches', n, 'res', res, 'onepass', one, '%.1f us per fwd+bwd' % (e0.elapsed_time(e1) * 1e3 / 50))
print('barrier error', dnn.onepass_error())
