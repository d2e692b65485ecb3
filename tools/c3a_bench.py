"""The regressor ResNets' layer1 convolutions (3x3 / stride 1, 64 -> 64 channels) on csrc/conv3x3a.hip against the LDS-tile kernel that ran
them before: time per launch from hipGraph replays, forward and data gradient.  python tools/c3a_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402
from c3s_bench import timeit                             # noqa: E402

L = _lib.lib()
for (B, H, W) in ((768, 16, 16), (32, 64, 64)):
    C = 64
    x = conv.nhwc_bf16(torch.randn(B, C, H, W, device='cuda'))
    w = torch.nn.Parameter(torch.randn(C, C, 3, 3, device='cuda') * 0.05)
    wp0, wp1, wa0, wa1 = conv.pack_weight(w, 1, 0), conv.pack_weight(w, 1, 1), conv.pack_weight(w, 1, 0, 16), conv.pack_weight(w, 1, 1, 16)
    sums = torch.zeros(L.danet_bn_ws_floats(C), device='cuda')
    flops = 2.0 * B * H * W * C * C * 9
    rec = {'shape': [B, H, W, C]}
    for name, fn in (('tile_fwd', lambda: conv._conv_fwd_raw(x, wp0, None, B, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1, False, False, False, sums)),
                     ('rowtile_fwd', lambda: conv._conv3x3a_raw(x, wa0, B, H, W, False, sums)),
                     ('tile_dgrad', lambda: conv._conv_fwd_raw(x, wp1, None, B, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1, True, False, False)),
                     ('rowtile_dgrad', lambda: conv._conv3x3a_raw(x, wa1, B, H, W, True))):
        t = timeit(fn, iters=10)
        rec[name] = {'us': round(t * 1e6, 1), 'frac': round(flops / t / 2.5e15, 4)}
    print(json.dumps(rec))
