"""Per-shape timing of the MFMA conv kernels (fwd / dgrad / wgrad) vs MIOpen through torch
(bf16 channels_last), B=32.  Prints TFLOP/s and the fraction of the 2.5 PF dense bf16 peak."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402
from danet_densepose2smpl_amd._lib import ptr, stream   # noqa: E402

PEAK = 2.5e15
SHAPES = [
    (48, 48, 3, 1, 1, 1, 64, 64), (96, 96, 3, 1, 1, 1, 32, 32), (192, 192, 3, 1, 1, 1, 16, 16),
    (384, 384, 3, 1, 1, 1, 8, 8), (64, 64, 3, 1, 1, 1, 64, 64), (256, 48, 3, 1, 1, 1, 64, 64), (64, 64, 3, 1, 1, 1, 16, 16),
    (64, 256, 1, 1, 0, 1, 64, 64), (48, 96, 3, 2, 1, 1, 64, 64), (48 * 24, 21 * 24, 3, 1, 1, 24, 64, 64),
]


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = _lib.lib()
    only = sys.argv[2] if len(sys.argv) > 2 else ''
    sel = os.environ.get('DANET_MB_SHAPES')
    shapes = SHAPES if not sel else [SHAPES[int(i)] for i in sel.split(',')]
    if os.environ.get('DANET_MB_SHAPE'):               # "Cin,Cout,k,stride,pad,groups,H,W"
        shapes = [tuple(int(v) for v in os.environ['DANET_MB_SHAPE'].split(','))]
    for (Cin, Cout, k, s, p, g, H, W) in shapes:
        OH, OW = conv.conv_out_size(H, k, s, p, 1), conv.conv_out_size(W, k, s, p, 1)
        flops = 2.0 * B * OH * OW * Cout * (Cin // g) * k * k
        x = conv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))
        w = torch.nn.Parameter(torch.randn(Cout, Cin // g, k, k, device='cuda') * 0.05)      # Parameter: packed once (cached)
        gy = conv.nhwc_bf16(torch.randn(B, Cout, OH, OW, device='cuda'))
        wp0, wp1 = conv.pack_weight(w, g, 0), conv.pack_weight(w, g, 1)
        res = {'shape': [Cin, Cout, k, s, p, g, H, W], 'B': B, 'GFLOP': flops / 1e9}
        t = timeit(lambda: conv._conv_fwd_raw(x, wp0, None, B, H, W, Cin, OH, OW, Cout, k, k, s, p, 1, g, False, False, False))
        res['fwd_us'] = t * 1e6; res['fwd_TF'] = flops / t / 1e12
        t = timeit(lambda: conv._conv_fwd_raw(gy, wp1, None, B, OH, OW, Cout, H, W, Cin, k, k, s, p, 1, g, True, False, False))
        res['dgrad_us'] = t * 1e6; res['dgrad_TF'] = flops / t / 1e12
        gw = torch.empty_like(w.data)
        nws = L.danet_conv_wgrad_ws_floats(Cout, Cin // g, k, k)
        ws = torch.empty(nws, device='cuda')
        xp, gyp = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
        t = timeit(lambda: L.danet_conv_wgrad(ptr(xp), ptr(gyp), ptr(gw), ptr(ws), nws, B, H, W, Cin, OH, OW, Cout, k, k, s, p, 1, g, 0.0, 0, stream()))
        res['wgrad_us'] = t * 1e6; res['wgrad_TF'] = flops / t / 1e12
        if L.danet_conv_wgrad_rows_ok(B, H, W, Cin, OH, OW, Cout, k, k, s, p, 1, g):
            nr = L.danet_conv_wgrad_rows_ws_floats(B, OH, OW, Cin, Cout, k, k, g)
            wsr = torch.empty(nr, device='cuda')
            gwr = torch.empty_like(w.data)
            t = timeit(lambda: L.danet_conv_wgrad_rows(ptr(xp), ptr(gyp), ptr(gwr), ptr(wsr), nr, B, H, W, Cin, OH, OW, Cout, k, k, s, p, g, 0.0, stream()))
            res['wgrad_rows_us'] = t * 1e6; res['wgrad_rows_TF'] = flops / t / 1e12
            res['wgrad_rows_vs_old_maxrel'] = float((gwr - gw).abs().max() / (gw.abs().max() + 1e-9))
        if L.danet_conv_wgrad3x3_ok(H, W, Cin, Cout, k, k, s, p, 1, g):
            n3 = L.danet_conv_wgrad3x3_ws_floats(B, H, W, Cin, Cout, g, s)
            ws3 = torch.empty(n3, device='cuda')
            gw3 = torch.empty_like(w.data)
            t = timeit(lambda: L.danet_conv_wgrad3x3(ptr(xp), ptr(gyp), ptr(gw3), ptr(ws3), n3, B, H, W, Cin, Cout, g, s, 0.0, 0, stream()))
            res['wgrad3_us'] = t * 1e6; res['wgrad3_TF'] = flops / t / 1e12
            res['wgrad3_vs_old_maxrel'] = float((gw3 - gw).abs().max() / (gw.abs().max() + 1e-9))
        if only != 'nomiopen':
            xb = x.detach().clone().requires_grad_(True)
            wb = w.detach().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            t = timeit(lambda: F.conv2d(xb, wb, None, s, p, 1, g))
            res['miopen_fwd_us'] = t * 1e6
            y = F.conv2d(xb, wb, None, s, p, 1, g)

            def fb():
                yy = F.conv2d(xb, wb, None, s, p, 1, g)
                yy.backward(gy)
                xb.grad = None; wb.grad = None
            t2 = timeit(fb)
            res['miopen_fwdbwd_us'] = t2 * 1e6
        res['ours_fwdbwd_us'] = res['fwd_us'] + res['dgrad_us'] + res['wgrad_us']
        res['frac_peak_fwd'] = res['fwd_TF'] * 1e12 / PEAK
        print(json.dumps({k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in res.items()}))


if __name__ == '__main__':
    main()
