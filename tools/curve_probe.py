"""Loss curves of the bf16 and the fp32 train step from the same initial weights on one fixed batch (exploration for
tests/test_gpu_fp32.py::test_bf16_training_curve_tracks_fp32).  usage: python tools/curve_probe.py [steps] [lr] [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv                                                     # noqa: E402
from danet_densepose2smpl_amd.config import cfg                                               # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options      # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, cfg.DANET.PARTDROP_RATE = 128, 32, 0.
    cfg.DANET.STN_CENTER_JITTER = cfg.DANET.STN_SCALE_JITTER = 0.
    dev = torch.device('cuda')
    curves = {}
    for mode in ('bf16', 'bf16-again', 'fp32'):
        torch.manual_seed(0)
        tr = Trainer(default_options(B), device=dev, distributed=False, lr=lr)
        batch = synthetic_in_dict(tr.model, B, dev, seed=1)
        tot = []
        for _ in range(steps):
            if mode == 'fp32':
                with conv.precision('fp32'):
                    _, ls = tr.train_step(batch)
            else:
                _, ls = tr.train_step(batch)
            tot.append(round(sum(float(v.sum()) for v in ls.values()), 4))
        curves[mode] = tot
        del tr
    print(json.dumps(curves))


main()
