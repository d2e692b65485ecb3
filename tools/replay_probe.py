"""After each of `n` hipGraph replays (and `n` eager steps first) of the train step on one batch: which gradients / losses /
parameters hold non-finite values.  usage: python tools/replay_probe.py [B] [size] [n]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg                                          # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-30
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, cfg.DANET.PARTDROP_RATE = size, size // 4, 0.
    cfg.DANET.STN_CENTER_JITTER = cfg.DANET.STN_SCALE_JITTER = 0.
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(B), device=dev, distributed=False, lr=lr)
    batch = synthetic_in_dict(tr.model, B, dev, seed=1)

    def bad(tag, losses):
        torch.cuda.synchronize()
        g = [(k, float(p.grad.abs().max())) for k, p in tr.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        w = [k for k, p in tr.model.named_parameters() if not torch.isfinite(p).all()]
        lo = [k for k, v in losses.items() if not torch.isfinite(v).all()]
        print(json.dumps({'step': tag, 'bad_grads': g[:8], 'n_bad_grads': len(g), 'bad_params': w[:8], 'bad_losses': lo,
                          'total': float(sum(v.float().sum() for v in losses.values()))}))
    for i in range(n):
        bad('eager%d' % i, tr.train_step(batch)[1])
    tr.capture(batch, warmup=1)
    for i in range(n):
        bad('graph%d' % i, tr.train_step_graphed()[1])


main()
from danet_densepose2smpl_amd import conv as _c     # noqa: E402
print(json.dumps({'arena_refused_during_capture': _c.ARENA.refused[:10], 'n': len(_c.ARENA.refused), 'high': _c.ARENA.high, 'zeroed': _c.ARENA.zeroed}))
