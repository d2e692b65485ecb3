"""The lockstep launches of conv3x3_stream_kernel (2, 3 and 4 HRNet branches, B = 32, forward with statistics and data gradient) with the
launch's tiles dealt over all workgroups (rounds 2-4) against workgroups dealt to the problems in proportion to their work (round 5,
s3_assign), for several values of the per-tile fixed cost of its model.  Outputs must be bit-identical.  One JSON line per measurement."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402
from c3s_bench import timeit, PEAK                        # noqa: E402

L = _lib.lib()
B = 32
ALL = ((48, 64), (96, 32), (192, 16), (384, 8))
for nb in (2, 3, 4):
    shapes = ALL[:nb]
    xs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in shapes]
    wps = [(conv.pack_weight(torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05), 1, 0),) for c, s in shapes]
    wts = [conv.pack_weight(torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05), 1, 1) for c, s in shapes]
    ys = [torch.empty_like(x) for x in xs]
    sums = [torch.zeros(L.danet_bn_ws_floats(c), device='cuda') for c, s in shapes]
    flops = sum(2.0 * B * s * s * c * c * 9 for c, s in shapes)
    for transposed in (False, True):
        jobs = (_lib.ConvJob * nb)()
        for j, x, wp, wt, y, (c, s), sm in zip(jobs, xs, wps, wts, ys, shapes, sums):
            conv._conv_job(j, x, wt if transposed else wp[0], y, (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1), transposed, None if transposed else sm)
        assert L.danet_conv_forward_multi_kernel(ctypes.addressof(jobs), nb) == 3

        def multi():
            conv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), nb, _lib.stream()), 'multi')
        ref = None
        for bal, cost in ((0, 8), (1, 0), (1, 4), (1, 8), (1, 16), (1, 32), (0, 8)):
            with _lib.knobs(c3s_balance=bal, c3s_tile_cost=cost):
                for y in ys:
                    y.zero_()
                multi()
                torch.cuda.synchronize()
                out = torch.cat([y.reshape(-1).float() for y in ys])
                if ref is None:
                    ref = out.clone()
                t = timeit(multi)
            print(json.dumps({'branches': nb, 'pass': 'dgrad' if transposed else 'fwd+stats', 'balance': bal, 'tile_cost': cost, 'us': round(t * 1e6, 2),
                              'frac': round(flops / t / PEAK, 4), 'identical': bool(torch.equal(out, ref))}), flush=True)
