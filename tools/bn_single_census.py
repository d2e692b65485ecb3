"""The single-tensor BatchNorm launches of one train step (nn.BatchNormActFunction: everything the lockstep multi-problem launches do
not cover) by input shape, with the kernel that produced the input and whether its epilogue accumulated the statistics.
usage: python tools/bn_single_census.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import nn as dnn, conv as dconv                                  # noqa: E402
from danet_densepose2smpl_amd.config import cfg                                               # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options      # noqa: E402


def main():
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE = 256, 64
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(32), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 32, dev, seed=3)
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    seen = collections.Counter()
    convs = []
    orig_fwd = dnn.BatchNormActFunction.forward
    orig_raw = dconv._conv_fwd_raw

    def raw(x, wp, bias, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, *a, **k):
        y = orig_raw(x, wp, bias, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, *a, **k)
        if not transposed:
            kid = dconv._lib.lib().danet_conv_forward_kernel(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, 0, 1)
            convs.append(((B, Cout, OH, OW), 'k%dx%d s%d g%d cin%d id%d' % (R, S, stride, groups, Cin, kid)))
        return y

    def fwd(ctx, x, *a, **k):
        shp = tuple(x.shape)
        prod = next((c[1] for c in reversed(convs[-3:]) if c[0] == shp), '?')
        seen[(shp, prod, getattr(x, '_bn_sums', None) is not None)] += 1
        return orig_fwd(ctx, x, *a, **k)
    dnn.BatchNormActFunction.forward = staticmethod(fwd)
    dconv._conv_fwd_raw = raw
    try:
        tr.train_step(batch)
    finally:
        dnn.BatchNormActFunction.forward = staticmethod(orig_fwd)
        dconv._conv_fwd_raw = orig_raw
    torch.cuda.synchronize()
    tot = 0
    for (shp, prod, fused), n in sorted(seen.items(), key=lambda kv: -kv[1] * kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3]):
        mb = shp[0] * shp[1] * shp[2] * shp[3] * 2 / 2**20
        tot += n
        print('%3d x  %-22s %7.2f MB  stats-in-conv=%d  from %s' % (n, shp, mb, fused, prod))
    print('single BatchNorm launches per step:', tot)


main()
