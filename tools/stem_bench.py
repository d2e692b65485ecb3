"""The regressors' 7x7 / stride-2 stem over the 768 part crops (the step's largest single layer: 315 GFLOP forward, the same again
for the data gradient): time per launch from hipGraph replays.  python tools/stem_bench.py   (DANET_CONV_MT8=1 for 128-pixel wave tiles)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402
from c3s_bench import timeit                             # noqa: E402

L = _lib.lib()
B, C, H = 768, 64, 64
x = conv.nhwc_bf16(torch.randn(B, C, H, H, device='cuda'))
w = torch.nn.Parameter(torch.randn(C, C, 7, 7, device='cuda') * 0.02)
gy = conv.nhwc_bf16(torch.randn(B, C, H // 2, H // 2, device='cuda'))
wp0, wp1 = conv.pack_weight(w, 1, 0), conv.pack_weight(w, 1, 1)
flops = 2.0 * B * (H // 2) ** 2 * C * C * 49


wp16 = conv.pack_weight(w, 1, 0, 16)
USE_STEM = bool(L.danet_conv_stem_ok(B, H, H, C, H // 2, H // 2, C, 7, 7, 2, 3, 1, 1))


def fwd():
    if USE_STEM:
        return conv._conv_stem_raw(x, wp16, B, H, H, C, H // 2, H // 2, C)
    return conv._conv_fwd_raw(x, wp0, None, B, H, H, C, H // 2, H // 2, C, 7, 7, 2, 3, 1, 1, False, False, False)


def dgrad():
    return conv._conv_fwd_raw(gy, wp1, None, B, H // 2, H // 2, C, H, H, C, 7, 7, 2, 3, 1, 1, True, False, False)


wp16t = conv.pack_weight(w, 1, 1, 16)
bn_x = conv.nhwc_bf16(torch.randn(B, C, H, H, device='cuda'))
bn_y = conv.nhwc_bf16(torch.randn(B, C, H, H, device='cuda'))
saved = torch.cat([torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5])
red = torch.zeros(L.danet_bn_ws_floats(C), device='cuda')
USE_DG = bool(L.danet_conv_stem_dgrad_ok(B, H, H, C, H // 2, H // 2, C, 7, 7, 2, 3, 1, 1))


def dgrad_tile(bn=False):
    return conv._conv_stem_dgrad_raw(gy, wp16t, B, H, H, C, H // 2, H // 2, C, (bn_x, bn_y, saved, red, 0) if bn else None)


def dgrad_gather_bn():
    return conv._conv_fwd_raw(gy, wp1, None, B, H // 2, H // 2, C, H, H, C, 7, 7, 2, 3, 1, 1, True, False, False, None, (bn_x, bn_y, saved, red, 0))


yr = torch.nn.functional.conv2d(x.float()[:8], w.detach().bfloat16().float(), None, 2, 3)
y = fwd()
err = float((y[:8].float() - yr).abs().max() / yr.abs().max())
tf, tg = timeit(fwd, iters=5), timeit(dgrad, iters=5)
if USE_DG:
    gr = torch.nn.functional.conv_transpose2d(gy[:8].float(), w.detach().bfloat16().float(), None, 2, 3, 1)
    gt = dgrad_tile()
    err_d = float((gt[:8].float() - gr).abs().max() / gr.abs().max())
    td, tdb, tgb = timeit(dgrad_tile, iters=5), timeit(lambda: dgrad_tile(True), iters=5), timeit(dgrad_gather_bn, iters=5)
    print(json.dumps({'stem_dgrad': [B, C, H], 'tile_us': round(td * 1e6, 1), 'tile_frac': round(flops / td / 2.5e15, 4), 'tile_bn_us': round(tdb * 1e6, 1),
                      'gather_us': round(tg * 1e6, 1), 'gather_bn_us': round(tgb * 1e6, 1), 'err_dgrad': round(err_d, 5)}))
print(json.dumps({'stem': [B, C, H], 'lds_tile_kernel': USE_STEM, 'kernel_fwd': L.danet_conv_forward_kernel(B, H, H, C, H // 2, H // 2, C, 7, 7, 2, 3, 1, 1, 0, 0),
                  'fwd_us': round(tf * 1e6, 1), 'fwd_frac': round(flops / tf / 2.5e15, 4), 'dgrad_us': round(tg * 1e6, 1), 'dgrad_frac': round(flops / tg / 2.5e15, 4),
                  'err_fwd': round(err, 5)}))

# the 24-group partial-IUV head (3x3, 24 x (48 -> 24 padded) channels) at the step's size: streamed kernel vs gather kernel
G, Cg, Ng, Bh, Hh = 24, 48, 24, 32, 64
xg = conv.nhwc_bf16(torch.randn(Bh, G * Cg, Hh, Hh, device='cuda'))
wg = torch.nn.Parameter(torch.randn(G * Ng, Cg, 3, 3, device='cuda') * 0.05)
wpg = conv.pack_weight(wg, G, 0)
sums = torch.zeros(L.danet_bn_ws_floats(G * Ng), device='cuda')
fl = 2.0 * Bh * Hh * Hh * G * Ng * Cg * 9


def grouped(st=None):
    return conv._conv_fwd_raw(xg, wpg, None, Bh, Hh, Hh, G * Cg, Hh, Hh, G * Ng, 3, 3, 1, 1, 1, G, False, False, False, st)


rec = {'grouped_head': [Bh, G * Cg, G * Ng, Hh]}
for on in (0, 1):
    prev = L.danet_conv3x3_stream_set(on, -1, -1, -1)
    kid = L.danet_conv_forward_kernel(Bh, Hh, Hh, G * Cg, Hh, Hh, G * Ng, 3, 3, 1, 1, 1, G, 0, 0)
    t, ts = timeit(grouped, iters=5), timeit(lambda: grouped(sums), iters=5)
    L.danet_conv3x3_stream_set(prev, -1, -1, -1)
    rec['stream' if on else 'gather'] = {'kernel': kid, 'us': round(t * 1e6, 1), 'us_stats': round(ts * 1e6, 1), 'frac': round(fl / t / 2.5e15, 4)}
print(json.dumps(rec))
