"""The WHOLE captured train step replayed N times (default 4000): the grid barriers' error word, the losses and a sample of the parameters
after every 500 steps.  Every barrier kernel of the step (one-pass BatchNorm backward, graph tail forward / backward) runs N times beside
the side streams' branches.  python tools/step_soak.py [N]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg                 # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options   # noqa: E402
from danet_densepose2smpl_amd import nn as dnn                                       # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
reset_cfg(); cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
dev = torch.device('cuda')
torch.manual_seed(1)
tr = Trainer(default_options(32), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, 32, dev, seed=1)
tr.train_step(batch); tr.train_step(batch)
tr.capture(batch)
torch.cuda.synchronize()
t0 = time.time()
checks = []
for i in range(1, N + 1):
    out = tr.train_step_graphed()
    if i % 500 == 0 or i == N:
        torch.cuda.synchronize()
        losses = out[1]
        finite = bool(all(torch.isfinite(v.float()).all() for v in losses.values())) and \
            bool(all(torch.isfinite(p).all() for p in list(tr.model.parameters())[::7]))
        word = [int(b[2]) for b in dnn._ONEPASS_BAR.values()]
        checks.append({'step': i, 'finite': finite, 'barrier_error_word': word, 'side_live': dnn.SIDE_LIVE})
        if not finite or any(word):
            break
el = time.time() - t0
print(json.dumps({'steps': checks[-1]['step'], 'ms_per_step': round(el / checks[-1]['step'] * 1e3, 3), 'fusion_gcn_tail': tr.fusion_counts.get('gcn_tail', 0),
                  'all_finite': all(c['finite'] for c in checks), 'any_barrier_error': any(any(c['barrier_error_word']) for c in checks), 'checks': checks}))
