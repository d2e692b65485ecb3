"""Timing of the LDS-tile 3x3 kernel (csrc/conv3x3.hip) against the gather kernel (csrc/conv_fast.hip) per shape,
register tiling and workgroup cap, plus the four-branch lockstep launch.  One JSON line per measurement."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402

PEAK = 2.5e15
SHAPES = [(48, 48, 64, 64, 32), (96, 96, 32, 32, 32), (192, 192, 16, 16, 32), (384, 384, 8, 8, 32),
          (64, 64, 64, 64, 32), (64, 64, 16, 16, 768), (128, 128, 8, 8, 768), (256, 256, 4, 4, 768)]


def timeit(fn, iters=40, warm=5):
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
    L = _lib.lib()
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    for (Cin, Cout, H, W, B) in SHAPES[:4] if quick else SHAPES:
        flops = 2.0 * B * H * W * Cout * Cin * 9
        x = conv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))
        w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05)
        gy = conv.nhwc_bf16(torch.randn(B, Cout, H, W, device='cuda'))
        wp0, wp1 = conv.pack_weight(w, 1, 0), conv.pack_weight(w, 1, 1)
        sums = torch.zeros(L.danet_bn_ws_floats(Cout), device='cuda')

        def fwd(st=None):
            return conv._conv_fwd_raw(x, wp0, None, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, False, False, False, st)

        def dgrad():
            return conv._conv_fwd_raw(gy, wp1, None, B, H, W, Cout, H, W, Cin, 3, 3, 1, 1, 1, 1, True, False, False)

        L.danet_conv3x3_set(0, 0, 0, 512, -1)
        y_ref, g_ref = fwd().float(), dgrad().float()
        base = {'shape': [Cin, Cout, H, W, B], 'GFLOP': round(flops / 1e9, 2)}
        t = timeit(fwd)
        print(json.dumps(dict(base, kernel='conv_fast', fwd_us=round(t * 1e6, 2), frac=round(flops / t / PEAK, 4))), flush=True)
        for (mt, kw) in [(0, 0), (8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (4, 4)]:
            L.danet_conv3x3_set(1, mt, kw, 512, -1)
            kid = L.danet_conv_forward_kernel(B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, 0, 0)
            if kid % 10 != 2:
                continue
            if (mt, kw) != (0, 0) and (kid // 1000, (kid // 10) % 10) != (mt, kw):
                continue
            err_f = float((fwd().float() - y_ref).abs().max() / y_ref.abs().max())
            err_g = float((dgrad().float() - g_ref).abs().max() / g_ref.abs().max())
            if os.environ.get('C3_PHASES'):
                nb = 1024
                dbg = torch.zeros(nb * 8, dtype=torch.int32, device='cuda')
                L.danet_conv3x3_set(1, mt, kw, 512, -1)
                for f_ in (fwd, lambda: fwd(sums)):
                    dbg.zero_()
                    L.danet_conv3x3_debug(dbg.data_ptr())
                    f_()
                    torch.cuda.synchronize()
                    L.danet_conv3x3_debug(None)
                    d = dbg.view(nb, 8).cpu().numpy().astype('int64')
                    d = d[d[:, 0] != 0]
                    ph = [float(((d[:, k + 1] - d[:, k]) & 0xffffffff).mean()) for k in range(6)]
                    print(json.dumps(dict(base, phases_cycles=dict(zip(['prologue', 'stage', 'kloop', 'ksplit', 'epilogue', 'flush'], [round(v) for v in ph])),
                                          tiling=[kid // 1000, (kid // 100) % 10, (kid // 10) % 10], wgs=int(d.shape[0]),
                                          total=round(float(((d[:, 6] - d[:, 0]) & 0xffffffff).mean())))), flush=True)
            for blocks in ((512,) if quick else (256, 512, 768, 1024)):
                L.danet_conv3x3_set(1, mt, kw, blocks, -1)
                tf = timeit(fwd)
                ts = timeit(lambda: fwd(sums))
                tg = timeit(dgrad)
                print(json.dumps(dict(base, kernel='conv3x3', tiling=[kid // 1000, (kid // 100) % 10, (kid // 10) % 10], forced=[mt, kw], blocks=blocks,
                                      fwd_us=round(tf * 1e6, 2), fwd_stats_us=round(ts * 1e6, 2), dgrad_us=round(tg * 1e6, 2),
                                      frac=round(flops / tf / PEAK, 4), err_fwd=round(err_f, 5), err_dgrad=round(err_g, 5))), flush=True)
        L.danet_conv3x3_set(1, 0, 0, 512, -1)

    # the four HRNet branches in one launch
    chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
    B = 32
    xs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in zip(chans, sizes)]
    ws = [torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05) for c in chans]
    wps = [conv.pack_weight(w, 1, 0) for w in ws]
    ys = [torch.empty_like(x) for x in xs]
    sums = [torch.zeros(L.danet_bn_ws_floats(c), device='cuda') for c in chans]
    jobs = (_lib.ConvJob * 4)()
    for j, x, wp, y, c, s, sm in zip(jobs, xs, wps, ys, chans, sizes, sums):
        conv._conv_job(j, x, wp, y, (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1), False, sm)
    flops = sum(2.0 * B * s * s * c * c * 9 for c, s in zip(chans, sizes))

    def multi():
        conv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), 4, _lib.stream()), 'multi')
    for on, blocks, want in ((0, 512, 0), (1, 512, 0), (1, 512, 256), (1, 512, 512), (1, 768, 0), (1, 768, 512), (1, 1024, 512)):
        L.danet_conv3x3_set(on, 0, 0, blocks, want)
        t = timeit(multi)
        print(json.dumps({'multi4': 'conv3x3' if on else 'conv_fast', 'blocks': blocks, 'want': want, 'us': round(t * 1e6, 2), 'GFLOP': round(flops / 1e9, 2),
                          'frac': round(flops / t / PEAK, 4)}), flush=True)
    L.danet_conv3x3_set(1, 0, 0, 512, 0)


if __name__ == '__main__':
    main()
