"""Run-to-run spread of the losses of the bench-configuration train step (same batch, same weights, lr ~ 0), eager, with selected
kernels switched off: tells the atomics' ordering noise from a race in a new kernel.  usage: python tools/noise_probe.py [runs]"""
import json
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import _lib
from danet_densepose2smpl_amd.config import cfg
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, cfg.DANET.PARTDROP_RATE = 256, 64, 0.
    cfg.DANET.STN_CENTER_JITTER = cfg.DANET.STN_SCALE_JITTER = 0.
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(32), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 32, dev, seed=3)
    L = _lib.lib()
    for tag, setup in (('all kernels', lambda: None), ('no stem kernel', lambda: L.danet_conv_stem_set(0)),
                       ('no stem, no pointwise', lambda: (L.danet_conv_pw_set(0), L.danet_conv_pw_wgrad_set(0)))):
        setup()
        vals = []
        for _ in range(runs):
            _, le = tr.train_step(batch)
            torch.cuda.synchronize()
            vals.append({k: float(v.sum()) for k, v in le.items()})
        spread = {k: round((max(v[k] for v in vals) - min(v[k] for v in vals)) / (abs(np.mean([v[k] for v in vals])) + 1e-9), 4) for k in vals[0]}
        print(json.dumps({'config': tag, 'runs': runs, 'rel_spread': dict(sorted(spread.items(), key=lambda kv: -kv[1])[:6])}))


main()
