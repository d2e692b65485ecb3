"""Per-layer conv timing of one eager DaNet train step (HIP events around every conv launch):
prints the (kernel, shape) groups with the largest total time.  python tools/layer_profile.py [B] [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv                                 # noqa: E402
from danet_densepose2smpl_amd.config import cfg_from_dict                 # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg_from_dict({'DANET.INIMG_SIZE': size, 'DANET.HEATMAP_SIZE': size // 4})
dev = torch.device('cuda:0')
torch.manual_seed(0)
tr = Trainer(default_options(B), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, B, dev, seed=1)
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
conv.PROFILER = conv.KernelProfiler()
tr.train_step(batch)
torch.cuda.synchronize()
rows = sorted(conv.PROFILER.by_shape().items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('conv launches %d, total %.2f ms' % (sum(v[0] for _, v in rows), tot * 1e3))
print('%-34s %-44s %5s %9s %9s %8s' % ('kernel', '(kind,B,H,W,Cin,Cout,k,stride,groups)', 'n', 'total_us', 'avg_us', 'TF/s'))
for (key, shape), (n, t, f) in rows[:60]:
    print('%-34s %-44s %5d %9.1f %9.1f %8.1f' % (key, shape, n, t * 1e6, t / n * 1e6, f / t / 1e12))
