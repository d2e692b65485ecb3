"""Where the time of a conv -> BatchNorm one-launch call goes (csrc/conv3x3s.hip conv3x3_stream_bn_kernel): per-workgroup clock stamps
of the debug buffer -- end of the convolution part, end of the grid barrier, end of the statistics phase, end of the tail --
for the HRNet branch sets at B = 32, beside the launch's duration with and without the tail.
usage: python tools/c3s_bn_phases.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, nn as dnn, _lib          # noqa: E402


def timeit(fn, iters=20, warm=3, reps=5):
    """Seconds per call on hipGraph replays of `iters` back-to-back calls, on a stream the one-launch path is allowed on."""
    st = torch.cuda.Stream()
    prev, dnn.ONEPASS_STREAM = dnn.ONEPASS_STREAM, st
    try:
        with torch.cuda.stream(st):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / (reps * iters)
    finally:
        dnn.ONEPASS_STREAM = prev


def main():
    L = _lib.lib()
    B = 32
    for chans, sizes in (((48, 96), (64, 32)), ((48, 96, 192), (64, 32, 16)), ((48, 96, 192, 384), (64, 32, 16, 8))):
        for with_res in (False, True):
            convs = [conv.Conv2d(c, c, 3, 1, 1, bias=False).cuda() for c in chans]
            bns = [dnn.BatchNorm2d(c).cuda() for c in chans]
            xs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')).requires_grad_(True) for c, s in zip(chans, sizes)]
            rs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in zip(chans, sizes)] if with_res else None       # (NHWC, as in the model)

            def run():
                return dnn.multi_conv_bn(convs, xs, bns, rs, relu=True)
            rec = {'branches': len(chans), 'residual': with_res}
            for fuse in (False, True):
                dnn.CONV_BN = fuse
                conv.FUSION.clear()
                run()
                rec['one_launch' if fuse else 'two_launches'] = dict(conv.FUSION)
                rec['us_fused' if fuse else 'us_two'] = round(timeit(run) * 1e6, 2)
            nb = 1024
            dbg = torch.zeros(nb * 16, dtype=torch.int32, device='cuda')
            L.danet_conv3x3_debug(dbg.data_ptr())
            run()
            torch.cuda.synchronize()
            L.danet_conv3x3_debug(None)
            d = dbg.view(nb, 16).cpu().numpy().astype('int64')
            d = d[d[:, 0] != 0]

            def span(a, b):                      # per-workgroup ticks of s_memtime between two stamps: [mean, max]
                v = (d[:, b] - d[:, a]) & 0xffffffff
                v = v[v < (1 << 30)]
                return [int(v.mean()), int(v.max())]
            rec['ticks'] = {'convolutions': span(0, 7), 'barrier': span(7, 8), 'statistics': span(8, 5), 'apply': span(5, 12), 'whole': span(0, 12)}
            rec['wgs'] = int(d.shape[0])
            print(json.dumps(rec), flush=True)


main()
