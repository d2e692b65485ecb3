"""fp32 MFMA convolutions (csrc/conv_f32m.hip) against torch's fp32 convolution and the direct verification kernels:
forward / data gradient / weight gradient errors and times.  One JSON line per shape."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv          # noqa: E402

SHAPES = [  # B, Cin, H, W, Cout, k, stride, pad, groups, bias
    (2, 3, 32, 32, 64, 3, 2, 1, 1, False), (2, 64, 16, 16, 64, 3, 2, 1, 1, False), (2, 64, 16, 16, 256, 1, 1, 0, 1, False),
    (2, 48, 16, 16, 48, 3, 1, 1, 1, False), (2, 48, 16, 16, 96, 3, 2, 1, 1, False), (2, 96, 8, 8, 48, 1, 1, 0, 1, False),
    (2, 21, 24, 24, 64, 7, 2, 3, 1, False), (2, 48, 14, 14, 25, 1, 1, 0, 1, True), (3, 40, 9, 11, 24, 3, 1, 1, 1, True),
    (2, 96, 8, 8, 48, 1, 1, 0, 24, True), (2, 384, 8, 8, 384, 3, 1, 1, 1, False), (2, 18, 10, 10, 30, 3, 1, 1, 1, False),
    (2, 32, 12, 12, 20, 3, 2, 1, 1, False), (2, 192, 8, 8, 84, 3, 1, 1, 4, True), (2, 48, 8, 8, 42, 1, 1, 0, 2, False),
]
BIG = [(32, 48, 64, 64, 48, 3, 1, 1, 1, False), (32, 96, 32, 32, 96, 3, 1, 1, 1, False), (32, 192, 16, 16, 192, 3, 1, 1, 1, False),
       (32, 384, 8, 8, 384, 3, 1, 1, 1, False), (32, 64, 64, 64, 256, 1, 1, 0, 1, False), (32, 256, 64, 64, 64, 1, 1, 0, 1, False),
       (32, 64, 64, 64, 64, 3, 1, 1, 1, False), (32, 3, 256, 256, 64, 3, 2, 1, 1, False), (32, 64, 128, 128, 64, 3, 2, 1, 1, False),
       (32, 48, 64, 64, 96, 3, 2, 1, 1, False), (32, 384, 8, 8, 48, 1, 1, 0, 1, False)]


def run(shape, timing):
    B, Cin, H, W, Cout, k, st, pad, groups, bias = shape
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device='cuda', requires_grad=True)
    w = torch.nn.Parameter(torch.randn(Cout, Cin // groups, k, k, device='cuda') * 0.1)
    b = torch.nn.Parameter(torch.randn(Cout, device='cuda')) if bias else None
    yr = F.conv2d(x.double(), w.double(), None if b is None else b.double(), st, pad, 1, groups)
    gy = torch.randn_like(yr)
    gxr, gwr = torch.autograd.grad(yr, (x, w), gy)
    rec = {'shape': list(shape)}
    for name, flag in (('mfma', True), ('direct', False)):
        conv.F32_MFMA = flag
        with conv.precision('fp32'):
            y = conv.conv2d(x, w, b, st, pad, 1, groups)
            gx, gw = torch.autograd.grad(y, (x, w), gy.float())

        def rel(a, r):
            return float((a.double() - r).abs().max() / r.abs().max())
        rec[name] = [float('%.2e' % rel(y, yr)), float('%.2e' % rel(gx, gxr)), float('%.2e' % rel(gw, gwr))]
        if timing and flag:
            with conv.precision('fp32'):
                for what in ('fwd', 'bwd'):
                    ts = []
                    for i in range(6):
                        xx = x.detach().requires_grad_(True)
                        if what == 'fwd':
                            torch.cuda.synchronize(); t0 = time.perf_counter()
                            y = conv.conv2d(xx, w, b, st, pad, 1, groups)
                            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                        else:
                            y = conv.conv2d(xx, w, b, st, pad, 1, groups)
                            torch.cuda.synchronize(); t0 = time.perf_counter()
                            torch.autograd.grad(y, (xx, w), gy.float())
                            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                    flops = 2.0 * yr.numel() * (Cin // groups) * k * k
                    rec[what + '_us'] = round(min(ts) * 1e6, 1)
                    rec[what + '_tf'] = round(flops * (1 if what == 'fwd' else 2) / min(ts) / 1e12, 1)
    conv.F32_MFMA = True
    print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    for s in SHAPES:
        run(s, False)
    if len(sys.argv) > 1 and sys.argv[1] == 'big':
        for s in BIG:
            run(s, True)
