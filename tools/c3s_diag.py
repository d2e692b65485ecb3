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
            L.danet_conv3x3_stream_set(1, 512, kw)
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
            rec['mfma'] = {'prologue': ph(0, 1), 'ringfill': ph(1, 2), 'kloop': ph(2, 3), 'ksplit': ph(3, 4), 'epilogue': ph(4, 5), 'flush+bar': ph(5, 6), 'total': ph(0, 7)}
            rec['loader'] = {'start_vs_mfma': ph(0, 8), 'setup': ph(8, 13), 'rows': ph(13, 9), 'wait': ph(9, 10), 'stage0_done': ph(10, 11), 'total': ph(8, 12)}
            # workgroups resident together on a CU: overlapping [start, end) intervals of the MFMA role per CU id
            res = []
            for cu in np.unique(d[:, 15]):
                w = d[d[:, 15] == cu]
                res.append(max(int(((w[:, 0] <= a) & (a < w[:, 7])).sum()) for a in w[:, 0]))
            rec['cus'] = len(res)
            rec['resident_per_cu'] = [min(res), round(float(np.mean(res)), 2), max(res)]
            for bl in (256, 512):
                L.danet_conv3x3_stream_set(1, bl, kw)
                rec['us_%d' % bl] = round(timeit(fwd) * 1e6, 2)
            print(json.dumps(rec), flush=True)
        L.danet_conv3x3_stream_set(1, 512, 0)


if __name__ == '__main__':
    main()
