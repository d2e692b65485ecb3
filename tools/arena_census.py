"""Who takes how much of the per-step zeroed accumulator arena (conv.ARENA) in the bench configuration.  usage: python tools/arena_census.py [B] [size]"""
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv as dconv                                       # noqa: E402
from danet_densepose2smpl_amd.config import cfg                                          # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, cfg.DANET.PARTDROP_RATE = size, size // 4, 0.
dev = torch.device('cuda')
tr = Trainer(default_options(B), device=dev, distributed=False, lr=1e-30)
batch = synthetic_in_dict(tr.model, B, dev, seed=1)
tr.train_step(batch)
use = collections.Counter()
cnt = collections.Counter()
orig = dconv.ARENA.alloc


def alloc(n):
    f = sys._getframe(1)
    key = '%s:%d' % (os.path.basename(f.f_code.co_filename), f.f_lineno)
    use[key] += (n + 15) // 16 * 16
    cnt[key] += 1
    return orig(n)


dconv.ARENA.alloc = alloc
tr.train_step(batch)
torch.cuda.synchronize()
print(json.dumps({'high_floats': dconv.ARENA.high, 'MB': round(dconv.ARENA.high * 4 / 1e6, 1),
                  'by_site_MB': {k: [round(v * 4 / 1e6, 2), cnt[k]] for k, v in use.most_common()}}, indent=1))
