"""Streamed 3x3 kernel (csrc/conv3x3s.hip) against the tile kernel (csrc/conv3x3.hip) and the gather kernel
(csrc/conv_fast.hip): max error and time per shape, forward (with / without fused statistics) and data gradient, plus
the four-branch lockstep launch.  One JSON line per measurement.  python tools/c3s_bench.py [quick]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402

PEAK = 2.5e15
SHAPES = [(48, 48, 64, 64, 32), (96, 96, 32, 32, 32), (192, 192, 16, 16, 32), (384, 384, 8, 8, 32),
          (48, 48, 56, 56, 2), (96, 96, 28, 28, 3), (192, 192, 14, 14, 4), (384, 384, 7, 7, 6), (48, 32, 20, 12, 3),
          (32, 32, 64, 64, 32), (16, 16, 8, 8, 2), (48, 24, 64, 64, 2), (80, 48, 16, 16, 5)]


def timeit(fn, iters=20, warm=3, reps=5):
    """Seconds per call, measured on hipGraph replays of `iters` back-to-back calls (eager launches are host-bound at ~13 us)."""
    st = torch.cuda.Stream()
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
    return e0.elapsed_time(e1) * 1e-3 / (iters * reps)


def main():
    L = _lib.lib()
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    blocks = [int(b) for b in os.environ.get('C3S_BLOCKS', '512').split(',')]
    only_multi = len(sys.argv) > 1 and sys.argv[1] == 'multi'
    for (Cin, Cout, H, W, B) in [] if only_multi else (SHAPES[:4] if quick else SHAPES):
        flops = 2.0 * B * H * W * Cout * Cin * 9
        x = conv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))
        w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05)
        gy = conv.nhwc_bf16(torch.randn(B, Cout, H, W, device='cuda'))
        wp0, wp1 = conv.pack_weight(w, 1, 0), conv.pack_weight(w, 1, 1)
        nws = L.danet_bn_ws_floats(Cout)

        def fwd(st=None):
            return conv._conv_fwd_raw(x, wp0, None, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, False, False, False, st)

        def dgrad():
            return conv._conv_fwd_raw(gy, wp1, None, B, H, W, Cout, H, W, Cin, 3, 3, 1, 1, 1, 1, True, False, False)

        L.danet_conv3x3_set(0, 0, 0, 512, -1)
        y_ref, g_ref = fwd().float(), dgrad().float()
        s_ref = torch.stack([y_ref.sum(dim=(0, 2, 3)), (y_ref * y_ref).sum(dim=(0, 2, 3))])
        base = {'shape': [Cin, Cout, H, W, B], 'GFLOP': round(flops / 1e9, 2)}
        for stream_on in (0, 1):
            L.danet_conv3x3_set(1, 0, 0, 512, -1)
            L.danet_conv3x3_stream_set(stream_on, 512, -1, -1)
            sums = torch.zeros(nws, device='cuda')
            y = fwd(sums).float()
            torch.cuda.synchronize()
            err_f = float((y - y_ref).abs().max() / y_ref.abs().max())
            err_g = float((dgrad().float() - g_ref).abs().max() / g_ref.abs().max())
            s = conv.bn_sums_total(sums, Cout)
            err_s = float(((s - s_ref).abs().max(dim=1)[0] / s_ref.abs().max(dim=1)[0]).max())
            for bl in blocks:
                L.danet_conv3x3_stream_set(stream_on, bl, -1, -1)
                tf, tg = timeit(fwd), timeit(dgrad)
                sums.zero_()
                ts = timeit(lambda: fwd(sums))
                print(json.dumps(dict(base, kernel='stream' if stream_on else 'tile', blocks=bl, fwd_us=round(tf * 1e6, 2), fwd_stats_us=round(ts * 1e6, 2),
                                      dgrad_us=round(tg * 1e6, 2), frac=round(flops / tf / PEAK, 4), err_fwd=round(err_f, 5), err_dgrad=round(err_g, 5),
                                      err_stats=round(err_s, 5))), flush=True)
        L.danet_conv3x3_stream_set(1, 512, 0, -1)

    # the four HRNet branches in one launch
    chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
    B = 32
    xs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in zip(chans, sizes)]
    ws = [torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05) for c in chans]
    wps = [conv.pack_weight(w, 1, 0) for w in ws]
    ys = [torch.empty_like(x) for x in xs]
    sums = [torch.zeros(L.danet_bn_ws_floats(c), device='cuda') for c in chans]
    flops = sum(2.0 * B * s * s * c * c * 9 for c, s in zip(chans, sizes))
    for with_stats in (0, 1):
        jobs = (_lib.ConvJob * 4)()
        for j, x, wp, y, c, s, sm in zip(jobs, xs, wps, ys, chans, sizes, sums):
            conv._conv_job(j, x, wp, y, (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1), False, sm if with_stats else None)

        def multi():
            conv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), 4, _lib.stream()), 'multi')
        L.danet_conv3x3_set(0, 0, 0, 512, 0)
        multi()
        refs = [y.float().clone() for y in ys]
        for name, c3, st in (('conv_fast', 0, 0), ('tile', 1, 0), ('stream', 1, 1)):
            L.danet_conv3x3_set(c3, 0, 0, 512, 0)
            L.danet_conv3x3_stream_set(st, 512, -1, -1)
            for y in ys:
                y.zero_()
            multi()
            torch.cuda.synchronize()
            err = max(float((y.float() - r).abs().max() / r.abs().max()) for y, r in zip(ys, refs))
            for want in ((0, 256, 512) if st else (0,)):
                L.danet_conv3x3_stream_set(st, 512, -1, want)
                t = timeit(multi)
                plans = [L.danet_conv3x3_stream_plan(B, s, s, c, c, 4) for c, s in zip(chans, sizes)] if st else None
                print(json.dumps({'multi4': name, 'stats': with_stats, 'want': want, 'plans': plans, 'us': round(t * 1e6, 2), 'GFLOP': round(flops / 1e9, 2),
                                  'frac': round(flops / t / PEAK, 4), 'err': round(err, 5)}), flush=True)
            L.danet_conv3x3_stream_set(st, 512, -1, 0)
    L.danet_conv3x3_set(1, 0, 0, 512, 0)
    L.danet_conv3x3_stream_set(1, 512, 0, -1)


if __name__ == '__main__':
    main()
