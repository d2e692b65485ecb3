"""Diagnostics of the streamed 3x3 kernel (csrc/conv3x3s.hip): correctness for every forced K split, per-phase timestamps
of the MFMA waves and of the loader wave, workgroup-cap sweep.  One JSON line per measurement."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402
from c3s_bench import timeit                              # noqa: E402

SHAPES = [(48, 48, 64, 64, 32), (96, 96, 32, 32, 32), (192, 192, 16, 16, 32), (384, 384, 8, 8, 32), (96, 96, 28, 28, 3), (192, 192, 14, 14, 4)]


def main():
    L = _lib.lib()
    for (Cin, Cout, H, W, B) in SHAPES:
        x = conv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))
        w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05)
        wp0 = conv.pack_weight(w, 1, 0)

        def fwd():
            return conv._conv_fwd_raw(x, wp0, None, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, False, False, False, None)
        L.danet_conv3x3_set(0, 0, 0, 512, -1)
        y_ref = fwd().float()
        L.danet_conv3x3_set(1, 0, 0, 512, -1)
        for kw in (1, 2, 4):
            L.danet_conv3x3_stream_set(1, 512, kw, -1)
            plan = L.danet_conv3x3_stream_plan(B, H, W, Cin, Cout, 1)
            if plan == 0:
                continue
            errs = []
            for _ in range(3):
                y = fwd().float()
                torch.cuda.synchronize()
                errs.append(float((y - y_ref).abs().max() / y_ref.abs().max()))
            rec = {'shape': [Cin, Cout, H, W, B], 'kw': kw, 'plan': plan, 'err': [round(e, 5) for e in errs]}
            nb = 1024
            dbg = torch.zeros(nb * 16, dtype=torch.int32, device='cuda')
            L.danet_conv3x3_debug(dbg.data_ptr())
            fwd()
            torch.cuda.synchronize()
            L.danet_conv3x3_debug(None)
            d = dbg.view(nb, 16).cpu().numpy().astype('int64')
            d = d[d[:, 0] != 0]

            def ph(a, b):
                return int(((d[:, b] - d[:, a]) & 0xffffffff).mean())
            rec['wgs'] = int(d.shape[0])
            rec['phases'] = {'issue0': ph(0, 9), 'wait0': ph(9, 10), 'desc': ph(10, 8), 'lanebase': ph(8, 11), 'fill': ph(11, 12), 'issue1': ph(12, 2), 'kloop': ph(2, 3), 'ksplit': ph(3, 4), 'epilogue': ph(4, 5), 'flush+bar': ph(5, 6), 'total': ph(0, 7)}
            # workgroups resident together on a CU: overlapping [start, end) intervals of the MFMA role per CU id
            res = []
            for cu in np.unique(d[:, 15]):
                w = d[d[:, 15] == cu]
                res.append(max(int(((w[:, 0] <= a) & (a < w[:, 7])).sum()) for a in w[:, 0]))
            rec['cus'] = len(res)
            rec['resident_per_cu'] = [min(res), round(float(np.mean(res)), 2), max(res)]
            for bl in (256, 512):
                L.danet_conv3x3_stream_set(1, bl, kw, -1)
                rec['us_%d' % bl] = round(timeit(fwd) * 1e6, 2)
            print(json.dumps(rec), flush=True)
        L.danet_conv3x3_stream_set(1, 512, 0, -1)


def multi_subsets():
    """Per-problem error of multi-problem launches over subsets of the four HRNet branches (isolates hand-over bugs between problems)."""
    import ctypes
    import itertools
    L = _lib.lib()
    chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
    B = 32
    xs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in zip(chans, sizes)]
    ws = [torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05) for c in chans]
    wps = [conv.pack_weight(w, 1, 0) for w in ws]
    ys = [torch.empty_like(x) for x in xs]
    refs = []
    L.danet_conv3x3_set(0, 0, 0, 512, 0)
    for x, wp, c, s in zip(xs, wps, chans, sizes):
        refs.append(conv._conv_fwd_raw(x, wp, None, B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1, False, False, False, None).float())
    L.danet_conv3x3_set(1, 0, 0, 512, 0)
    L.danet_conv3x3_stream_set(1, 512, 0, -1)
    for n in (1, 2, 3, 4):
        for sub in itertools.combinations(range(4), n):
            jobs = (_lib.ConvJob * n)()
            for j, i in zip(jobs, sub):
                c, s = chans[i], sizes[i]
                conv._conv_job(j, xs[i], wps[i], ys[i], (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1), False, None)
            errs = []
            for rep in range(2):
                for i in sub:
                    ys[i].zero_()
                conv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), n, _lib.stream()), 'multi')
                torch.cuda.synchronize()
                errs.append([round(float((ys[i].float() - refs[i]).abs().max() / refs[i].abs().max()), 4) for i in sub])
            plans = [L.danet_conv3x3_stream_plan(B, sizes[i], sizes[i], chans[i], chans[i], n) for i in sub]
            print(json.dumps({'multi_subset': [chans[i] for i in sub], 'plans': plans, 'err': errs}), flush=True)



def multi_phases():
    """Workgroup timelines of the four-branch launch: start / end stamps of every workgroup, phases of its first tile."""
    import ctypes
    L = _lib.lib()
    chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
    B = 32
    xs = [conv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in zip(chans, sizes)]
    wps = [conv.pack_weight(torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05), 1, 0) for c in chans]
    ys = [torch.empty_like(x) for x in xs]
    jobs = (_lib.ConvJob * 4)()
    for j, x, wp, y, c, s in zip(jobs, xs, wps, ys, chans, sizes):
        conv._conv_job(j, x, wp, y, (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1), False, None)
    L.danet_conv3x3_set(1, 0, 0, 512, 0)
    L.danet_conv3x3_stream_set(1, 512, -1, 0)

    def multi():
        conv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), 4, _lib.stream()), 'multi')
    for _ in range(3):
        multi()
    nb = 1024
    dbg = torch.zeros(nb * 16, dtype=torch.int32, device='cuda')
    L.danet_conv3x3_debug(dbg.data_ptr())
    multi()
    torch.cuda.synchronize()
    L.danet_conv3x3_debug(None)
    d = dbg.view(nb, 16).cpu().numpy().astype('int64')
    d = d[d[:, 0] != 0]
    t0 = d[:, 0].min()
    start = (d[:, 0] - t0) & 0xffffffff
    end = (d[:, 7] - t0) & 0xffffffff

    def q(a):
        return [int(v) for v in np.percentile(a, [0, 10, 50, 90, 100])]

    def ph(a, b):
        return q((d[:, b] - d[:, a]) & 0xffffffff)
    print(json.dumps({'multi4_wgs': int(d.shape[0]), 'life': q(end - start), 'sum_issue': q(d[:, 1]), 'sum_ksteps': q(d[:, 13]), 'sum_tile_end': q(d[:, 14]),
                      'first_tile': {'issue0': ph(0, 9), 'wait0': ph(9, 10), 'desc+lanebase+fill': ph(10, 12), 'issue1': ph(12, 2), 'kloop': ph(2, 3),
                                     'ksplit': ph(3, 4), 'epilogue': ph(4, 5), 'flush+bar': ph(5, 6)}, 'us': round(timeit(multi) * 1e6, 2)}), flush=True)

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'multi':
        multi_subsets()
    elif len(sys.argv) > 1 and sys.argv[1] == 'phases':
        multi_phases()
    else:
        main()
        multi_subsets()
