"""Debug/bench helper: run a few DaNet train steps and print losses + timing."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, cfg   # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg_from_dict({'DANET.INIMG_SIZE': size, 'DANET.HEATMAP_SIZE': size // 4})
dev = torch.device('cuda:0')
torch.manual_seed(0)
tr = Trainer(default_options(B), device=dev, distributed=False)
print('params', sum(p.numel() for p in tr.model.parameters()) / 1e6, 'M')
batch = synthetic_in_dict(tr.model, B, dev, seed=1)
for s in range(steps):
    torch.cuda.synchronize(); t0 = time.time()
    out, losses = tr.train_step(batch)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('step', s, '%.1f ms' % (dt * 1e3), {k: round(float(v.sum()), 4) for k, v in losses.items()})
print('max mem GB', torch.cuda.max_memory_allocated() / 2**30)

if os.environ.get('DANET_PMC_CALIB'):
    # known-size traffic for calibrating the PMC byte counters (tools/pmc_traffic.sh): 512 MiB read + 512 MiB written
    a = torch.ones(256 * 2**20, dtype=torch.bfloat16, device=dev)
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    b.copy_(a)
    torch.cuda.synchronize()
