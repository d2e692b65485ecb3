"""Pointwise kernel (csrc/conv_pw.hip) against the gather kernel (csrc/conv_fast.hip) on the step's 1x1 / stride-1 layers: time per
launch from hipGraph replays, forward and data gradient, with their HBM floors (X read once + Y written once at 6 TB/s).
One JSON line per measurement.  python tools/pw_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402
from c3s_bench import timeit                             # noqa: E402

SHAPES = [(64, 256, 64, 64, 32), (256, 64, 64, 64, 32), (24, 64, 64, 64, 768), (64, 24, 64, 64, 768), (48, 16, 64, 64, 32), (16, 48, 64, 64, 32)]


def main():
    L = _lib.lib()
    for (Cin, Cout, H, W, B) in SHAPES:
        x = conv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))
        w = torch.nn.Parameter(torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05)
        wp = conv.pack_weight(w, 1, 0)
        sums = torch.zeros(L.danet_bn_ws_floats(Cout), device='cuda')
        mb = B * H * W * (Cin + Cout) * 2 / 1e6

        def fwd(st=None):
            return conv._conv_fwd_raw(x, wp, None, B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, False, False, False, st)
        rec = {'shape': [Cin, Cout, H, W, B], 'MB': round(mb, 1), 'floor_us': round(mb / 6.0, 1)}
        for on in (0, 1):
            prev = L.danet_conv_pw_set(on)
            kid = L.danet_conv_forward_kernel(B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, 0, 0)
            y = fwd().float()
            t = timeit(fwd)
            ts = timeit(lambda: fwd(sums))
            L.danet_conv_pw_set(prev)
            rec['pw' if on else 'gather'] = {'kernel': kid, 'us': round(t * 1e6, 1), 'us_stats': round(ts * 1e6, 1), 'TBps': round(mb / t / 1e6, 2)}
            if on:
                rec['err'] = round(float((y - yref).abs().max() / yref.abs().max()), 5)
            else:
                yref = y
        # weight gradient: dY and X read once
        gy = conv.nhwc_bf16(torch.randn(B, Cout, H, W, device='cuda'))
        gw = torch.empty(Cout, Cin, 1, 1, device='cuda')
        nws = L.danet_conv_wgrad_ws_floats_for(B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1)
        ws = torch.zeros(nws, device='cuda')

        def wgrad():
            conv.check(L.danet_conv_wgrad(_lib.ptr(x.permute(0, 2, 3, 1)), _lib.ptr(gy.permute(0, 2, 3, 1)), _lib.ptr(gw), _lib.ptr(ws), nws,
                                          B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, 0.0, 0, _lib.stream()), 'wgrad')
        for on in (0, 1):
            prev = L.danet_conv_pw_wgrad_set(on)
            wgrad()
            g = gw.clone()
            t = timeit(wgrad)
            L.danet_conv_pw_wgrad_set(prev)
            rec['wgrad_pw' if on else 'wgrad_generic'] = {'us': round(t * 1e6, 1), 'TBps': round(mb / t / 1e6, 2)}
            if on:
                rec['wgrad_err'] = round(float((g - gref).abs().max() / gref.abs().max()), 6)
            else:
                gref = g
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
